#!/bin/bash
O=gpurun_out/r03_16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in dcn fpn; do
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$cfg -- python $GRAFT_REPO_ROOT/bench.py --train --$cfg --steps 5 --warmup 2 > /tmp/pt_$cfg.log 2>&1; echo "prof rc $?"
tail -1 /tmp/pt_$cfg.log | cut -c1-200
cp $(find /tmp/pt_$cfg -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/train_${cfg}_kernel_stats.csv
done
