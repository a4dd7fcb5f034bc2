#!/bin/bash
O=gpurun_out/r03_9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_relation_bwd.py tests/test_gpu_proposal_roi.py -q --tb=short 2>&1 | grep -E "nms_pair_pos|passed|failed|Error" | head -12
timeout 300 python bench.py --train --learn-nms --steps 10 --warmup 3 > $O/train.json 2> $O/train.err; python -c "
import json;d=json.loads([l for l in open('$O/train.json') if l.startswith('{')][0]);print('TRAIN', d['value'], d['ms_per_step'])"
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-train-line --no-batch-sweep > $O/bench.json 2>/dev/null; python -c "
import json;d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][0]);print('INFER', d['value'], d['ms_per_step'], d['kernels_ms']['relnet_roi_pool_fwd'])"
