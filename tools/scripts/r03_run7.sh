#!/bin/bash
O=gpurun_out/r03_7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_proposal_roi.py tests/test_gpu_train_step.py tests/test_gpu_bottleneck.py -q --tb=short > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -15
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python -c "
import json;d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][0]);print('INFER', d['value'], d['ms_per_step'], d['batch_sweep']['1'], d['batch_sweep']['8']); print(d['kernels_ms']); print('TRAIN', d['train']['value'], d['train']['ms_per_step'], d['train'].get('at_16_images_per_gpu')); print(d['roofline']); print(d['parity']['worst'])"
timeout 200 python bench.py --steps 20 --warmup 5 --stem hip3 --no-cpu-baseline --no-parity --no-train-line --no-batch-sweep --no-kernel-timing > $O/bench_hip3.json 2>/dev/null; python -c "
import json;d=json.loads([l for l in open('$O/bench_hip3.json') if l.startswith('{')][0]);print('INFER hip3 stem', d['value'], d['ms_per_step'])"
