#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_relation_bwd.py -x -q --tb=short 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --train --learn-nms --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train lnms 8:', round(d['value'],1), round(d['ms_per_step'],3))"
done
timeout 300 python bench.py --train --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train 8:', round(d['value'],1), round(d['ms_per_step'],3))"
timeout 300 python bench.py --train --learn-nms --batch 16 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train lnms 16:', round(d['value'],1), round(d['ms_per_step'],3))"
