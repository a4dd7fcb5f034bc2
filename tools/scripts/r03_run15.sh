#!/bin/bash
O=gpurun_out/r03_15; mkdir -p $O
timeout 300 python bench.py --train --learn-nms --steps 10 --warmup 3 > $O/train.json 2> $O/train.err; python -c "
import json;d=json.loads([l for l in open('$O/train.json') if l.startswith('{')][0]);print('TRAIN', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptrain -- python $GRAFT_REPO_ROOT/bench.py --train --learn-nms --steps 5 --warmup 2 > /tmp/ptrain.log 2>&1; echo "prof rc $?"
cp $(find /tmp/ptrain -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/train_kernel_stats.csv
