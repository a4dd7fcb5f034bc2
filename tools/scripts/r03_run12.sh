#!/bin/bash
# ping-pong ring schedule (tile 16): parity over all tile configs + per-shape timing at 54 images
mkdir -p gpurun_out/r03_12
timeout 900 python -m pytest tests/test_gpu_gemm_tiles.py -x -q --tb=short 2>&1 | tail -8
TILES=0,8,16 timeout 600 python tools/bench_tiles.py 54 2>&1 | tee gpurun_out/r03_12/tiles_b54.log
