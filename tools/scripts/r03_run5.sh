#!/bin/bash
O=gpurun_out/r03_5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --train --learn-nms --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "prof rc $?"
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/train_kernel_stats.csv; rm -rf $O/prof
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-train-line > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python -c "
import json;d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][0]);print('INFER', d['value'], d['ms_per_step'], d['batch_sweep'])"
