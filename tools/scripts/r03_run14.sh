#!/bin/bash
mkdir -p gpurun_out/r03_14
timeout 900 python -m pytest tests/test_gpu_gemm_tiles.py -x -q --tb=short 2>&1 | tail -5
for B in 1 8; do TILES=0,4,5,17,18 timeout 600 python tools/bench_tiles.py $B 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_14/tiles_b$B.log; done
