#!/bin/bash
O=gpurun_out/r03_4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_step.py tests/test_gpu_losses.py -q --tb=short -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -15
timeout 300 python tools/bench_wgrad.py 8 > $O/wgrad_b8.log 2>&1; tail -12 $O/wgrad_b8.log
TILES=0,8,4 timeout 200 python tools/bench_tiles.py 8 > $O/tiles_b8.log 2>&1; tail -3 $O/tiles_b8.log
TILES=0,4,5 timeout 200 python tools/bench_tiles.py 1 > $O/tiles_b1.log 2>&1; tail -3 $O/tiles_b1.log
timeout 300 python bench.py --train --learn-nms --steps 10 --warmup 3 > $O/train.json 2> $O/train.err; echo "train rc $?"; python -c "
import json;d=json.loads([l for l in open('$O/train.json') if l.startswith('{')][0]);print('TRAIN', d['value'], d['ms_per_step'])"; tail -3 $O/train.err
timeout 300 python bench.py --train --learn-nms --batch 16 --steps 6 --warmup 2 > $O/train16.json 2> $O/train16.err; python -c "
import json;d=json.loads([l for l in open('$O/train16.json') if l.startswith('{')][0]);print('TRAIN16', d['value'], d['ms_per_step'])"
