#!/bin/bash
F="--no-cpu-baseline --no-parity --no-train-line --no-batch-sweep --no-kernel-timing"
for b in 54 27 18 36 54; do python bench.py $F --batch $b 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', round(d['ms_per_step']/$b*54,3), 'ms per 54')"; done
