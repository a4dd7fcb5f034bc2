#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_train_ops.py -x -q --tb=short 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_train_step.py -x -q --tb=short -k "gradients" 2>&1 | tail -3
timeout 300 python tools/bench_wgrad.py 8 2>&1 | grep -v amdgpu | tail -6
for i in 1 2; do
timeout 300 python bench.py --train --learn-nms --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train lnms 8:', round(d['value'],1), round(d['ms_per_step'],3))"
done
