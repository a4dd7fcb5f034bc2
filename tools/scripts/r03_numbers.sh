#!/bin/bash
# round-3 numbers of every graph other than the default bench line (one JSON line each under gpurun_out/r03_numbers/)
O=gpurun_out/r03_numbers; mkdir -p $O
run() { name=$1; shift; timeout 400 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads([l for l in open('$O/$name.json') if l.startswith('{"metric')][-1])
    print('$name', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', d['config'].get('images_per_gpu_per_step'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
F="--no-cpu-baseline --no-parity --no-batch-sweep --no-train-line --no-kernel-timing"
run plain2fc $F --no-relation
run lnms27 $F --learn-nms --batch 27
run dcn27 $F --dcn --batch 27
run fpn8 $F --fpn --batch 8
run train8 --train --steps 10 --warmup 3
run train16 --train --batch 16 --steps 6 --warmup 2
run train_lnms8 --train --learn-nms --steps 10 --warmup 3
run train_lnms4 --train --learn-nms --batch 4 --steps 10 --warmup 3
run train_lnms16 --train --learn-nms --batch 16 --steps 6 --warmup 2
run train_dcn8 --train --dcn --steps 10 --warmup 3
run train_fpn2 --train --fpn --batch 2 --steps 10 --warmup 3
