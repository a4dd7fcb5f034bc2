#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_deform.py -x -q --tb=short 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_train_step.py -x -q --tb=short -k "dcn or deform" 2>&1 | tail -3
timeout 300 python bench.py --train --dcn --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-220
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_dcn -- python $GRAFT_REPO_ROOT/bench.py --train --dcn --steps 5 --warmup 2 > /tmp/pt_dcn.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_18; cp $(find /tmp/pt_dcn -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03_18/train_dcn_kernel_stats.csv
