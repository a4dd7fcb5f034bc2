#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_gemm_tiles.py -x -q --tb=short 2>&1 | tail -4
TILES=0,8,16 timeout 600 python tools/bench_tiles.py 54 2>&1 | grep -v amdgpu.ids | grep "3x3\|reduce\|conv_new\|totals"
timeout 300 python tools/bench_ablate.py 2>&1 | grep -v amdgpu.ids | grep "korder 1"
