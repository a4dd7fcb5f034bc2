#!/bin/bash
t() { timeout 400 python bench.py "$@" 2>&1 | grep -a "AssertionError: non-finite\|^{\"metric" | tail -1 | cut -c1-130; }
export RELNET_BENCH_ONE_DEVICE=1
echo "graph, lr 0.0005 (reference, unscaled)"; RELNET_BENCH_LR=0.0005 t --gpus 2 --train --learn-nms --steps 8 --warmup 3
echo "graph, default lr"; t --gpus 2 --train --learn-nms --steps 8 --warmup 3
