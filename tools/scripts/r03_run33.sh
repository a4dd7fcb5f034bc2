#!/bin/bash
t() { timeout 400 python bench.py "$@" 2>&1 | grep -a "AssertionError: non-finite\|^{\"metric" | tail -1 | cut -c1-100; }
export RELNET_BENCH_ONE_DEVICE=1 RELNET_BENCH_LR=0.00025
echo "graph, no sync"; t --gpus 2 --train --learn-nms --steps 8 --warmup 3
echo "graph, sync before update"; RELNET_BENCH_SYNC_EACH_STEP=1 t --gpus 2 --train --learn-nms --steps 8 --warmup 3
echo "eager, no sync"; t --gpus 2 --train --learn-nms --steps 8 --warmup 3 --no-graph
