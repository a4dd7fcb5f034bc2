#!/bin/bash
# round 4: the hand-scheduled k-loop (tile 18): parity on every tile-test shape, then per-layer A/B against tile 8
O=gpurun_out/r04_2; mkdir -p $O
RELNET_TEST_TILES=8,18 timeout 900 python -m pytest tests/test_gpu_gemm_tiles.py -x -q > $O/tiles.log 2>&1; echo "pytest rc=$?" >> $O/tiles.log
tail -15 $O/tiles.log
TILES=8,18,8,18 timeout 600 python tools/bench_tiles.py 54 > $O/bench_tiles_b54.txt 2>&1
cat $O/bench_tiles_b54.txt
