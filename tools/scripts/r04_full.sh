#!/bin/bash
# round 4: the whole GPU test suite, then rocprofv3 kernel stats of the default bench (54 images), the 1-image step and the training step
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_full; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
FL="--no-cpu-baseline --no-kernel-timing --no-parity --no-batch-sweep --no-train-line --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p54 -- python $R/bench.py $FL > /tmp/p54.log 2>&1
cp $(find /tmp/p54 -name "*kernel_stats.csv" | head -1) $O/bench_b54_kernel_stats.csv; tail -1 /tmp/p54.log | cut -c1-160
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $R/bench.py $FL --batch 1 --steps 50 > /tmp/p1.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/bench_b1_kernel_stats.csv; tail -1 /tmp/p1.log | cut -c1-160
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $R/bench.py --train --learn-nms --batch 8 --steps 10 --warmup 3 > /tmp/pt.log 2>&1
cp $(find /tmp/pt -name "*kernel_stats.csv" | head -1) $O/train_lnms_b8_kernel_stats.csv; tail -1 /tmp/pt.log | cut -c1-160
ls -la $O
