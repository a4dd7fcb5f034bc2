#!/bin/bash
O=gpurun_out/r04_9; mkdir -p $O
RELNET_TEST_TILES=20 timeout 900 python -m pytest tests/test_gpu_gemm_tiles.py -x -q > $O/tests.log 2>&1; tail -8 $O/tests.log
TILES=19,20,19,20 timeout 600 python tools/bench_tiles.py 54 > $O/bench_tiles_b54.txt 2>&1; grep "3x3\|totals" $O/bench_tiles_b54.txt
