#!/bin/bash
F="--no-cpu-baseline --no-parity --no-train-line --no-batch-sweep --no-kernel-timing"
for k in 0 1 0 1; do RELNET_GEMM_KORDER=$k python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('korder $k', round(d['value'],1), round(d['ms_per_step'],3))"; done
