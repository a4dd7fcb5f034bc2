#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_ops.py tests/test_gpu_dataset.py -x -q --tb=short 2>&1 | tail -5
timeout 300 python bench.py --train --learn-nms --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python bench.py --train --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-200
