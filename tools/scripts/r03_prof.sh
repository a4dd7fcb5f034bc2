#!/bin/bash
# round-3 profiles: kernel stats of the default bench (54 images) and of the 1-image step, PMC passes on the relation kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_prof; mkdir -p $O
FL="--no-cpu-baseline --no-kernel-timing --no-parity --no-batch-sweep --no-train-line"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p54 -- python $R/bench.py $FL > /tmp/p54.log 2>&1
cp $(find /tmp/p54 -name "*kernel_stats.csv" | head -1) $O/bench_b54_kernel_stats.csv
tail -1 /tmp/p54.log | cut -c1-160
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $R/bench.py $FL --batch 1 --steps 50 > /tmp/p1.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/bench_b1_kernel_stats.csv
tail -1 /tmp/p1.log | cut -c1-160
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY"; do
  tag=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pa_$tag -- python $R/tools/attn_only.py 54 6 > /tmp/pa.log 2>&1
  tail -1 /tmp/pa.log
done
python $R/tools/pmc_collect.py $O/attention_pmc_raw.json /tmp/pa_FETCH_SIZE /tmp/pa_WRITE_SIZE /tmp/pa_SQ_VALU_MFMA_BUSY_CYCLES
ls -la $O
