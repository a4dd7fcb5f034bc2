#!/bin/bash
O=gpurun_out/r03_3; mkdir -p $O
timeout 300 python tools/bench_wgrad.py 8 > $O/wgrad_b8.log 2>&1; tail -25 $O/wgrad_b8.log
TILES=0,8,2,3,4 timeout 300 python tools/bench_tiles.py 8 > $O/tiles_b8.log 2>&1; tail -20 $O/tiles_b8.log
TILES=0,8,2,3,4,5 timeout 300 python tools/bench_tiles.py 1 > $O/tiles_b1.log 2>&1; tail -20 $O/tiles_b1.log
