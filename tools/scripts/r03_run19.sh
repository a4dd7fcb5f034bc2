#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_plain -- python $GRAFT_REPO_ROOT/bench.py --train --steps 5 --warmup 2 > /tmp/pt_plain.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_19; cp $(find /tmp/pt_plain -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03_19/train_plain_kernel_stats.csv
