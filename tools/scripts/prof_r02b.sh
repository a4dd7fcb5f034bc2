cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02p
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $R/bench.py --no-cpu-baseline --no-kernel-timing --no-parity --no-batch-sweep --no-train-line > /tmp/p_bench.log 2>&1
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02p/bench_b54_kernel_stats.csv
cp $(find /tmp/p_bench -name "*domain_stats.csv" | head -1) $R/gpurun_out/r02p/bench_b54_domain_stats.csv 2>/dev/null
tail -1 /tmp/p_bench.log | cut -c1-200
