#!/bin/bash
mkdir -p gpurun_out/r03_24
timeout 900 python -m pytest tests/test_gpu_gemm_tiles.py -x -q --tb=short 2>&1 | tail -8
TILES=0,8,17 timeout 600 python tools/bench_tiles.py 54 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_24/tiles_b54.log
