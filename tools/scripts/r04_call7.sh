#!/bin/bash
O=gpurun_out/r04_7; mkdir -p $O
timeout 600 python tools/fill_probe.py > $O/fill_probe.txt 2>&1; cat $O/fill_probe.txt
F="--no-cpu-baseline --no-train-line --no-other-configs --no-parity --no-batch-sweep"
for i in 1 2; do python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('t19 default', round(d['value'],1), round(d['ms_per_step'],3))"; done
