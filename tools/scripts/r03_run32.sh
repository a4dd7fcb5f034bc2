#!/bin/bash
t() { timeout 400 python bench.py "$@" 2>&1 | grep -a "AssertionError: non-finite\|^{\"metric" | tail -1 | cut -c1-120; }
echo "world 1, B 8, lr 0.0005 (old default)"; RELNET_BENCH_LR=0.0005 t --train --learn-nms --steps 8 --warmup 3
echo "world 1, B 8, lr 0.001 (gradient scale of 2 summed ranks)"; RELNET_BENCH_LR=0.001 t --train --learn-nms --steps 8 --warmup 3
echo "world 1, B 16, lr 0.0005"; RELNET_BENCH_LR=0.0005 t --train --learn-nms --batch 16 --steps 8 --warmup 3
export RELNET_BENCH_ONE_DEVICE=1
echo "world 2 (one device), B 8, lr 0.0005"; RELNET_BENCH_LR=0.0005 t --gpus 2 --train --learn-nms --steps 8 --warmup 3
echo "world 2 (one device), B 8, lr 0.00025"; RELNET_BENCH_LR=0.00025 t --gpus 2 --train --learn-nms --steps 8 --warmup 3
