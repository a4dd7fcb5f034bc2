#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_two_ranks.py -x -q --tb=short 2>&1 | tail -4
t() { timeout 300 python bench.py --train --learn-nms --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['weights_finite_on_all_ranks'])"; }
echo "overlap on"; t; echo "overlap off"; RELNET_TRAIN_OVERLAP=0 t; echo "overlap on"; t
