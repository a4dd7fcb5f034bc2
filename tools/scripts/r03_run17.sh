#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_deform.py -x -q --tb=short 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_train_step.py -x -q --tb=short -k "dcn or deform" 2>&1 | tail -5
timeout 300 python bench.py --train --dcn --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-220
