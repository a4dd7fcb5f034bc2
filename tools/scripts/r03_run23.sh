#!/bin/bash
F="--no-cpu-baseline --no-parity --no-train-line --no-batch-sweep --no-kernel-timing"
for m in 4 1 4 1; do for b in 1 2; do RELNET_OVERLAP_MIN_IMAGES=$m python bench.py $F --batch $b --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap_min $m batch $b', round(d['value'],1), round(d['ms_per_step'],3))"; done; done
