cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02p
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $R/bench.py --no-cpu-baseline --no-kernel-timing --no-parity --no-batch-sweep --no-train-line > /tmp/p_bench.log 2>&1
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02p/bench_b54_kernel_stats.csv
cp $(find /tmp/p_bench -name "*domain_stats.csv" | head -1) $R/gpurun_out/r02p/bench_b54_domain_stats.csv 2>/dev/null
tail -1 /tmp/p_bench.log | cut -c1-200
for T in 1 8; do
  for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $C | cut -d' ' -f1)
    TILE=$T rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_pmc_${T}_${tag} -- python $R/tools/conv_only.py 54 3 > /tmp/pmc.log 2>&1
    f=$(find /tmp/p_pmc_${T}_${tag} -name "*counter_collection.csv" | head -1)
    cp $f $R/gpurun_out/r02p/pmc_tile${T}_${tag}.csv
    k=$(find /tmp/p_pmc_${T}_${tag} -name "*kernel_trace.csv" | head -1)
    cp $k $R/gpurun_out/r02p/trace_tile${T}_${tag}.csv
  done
done
ls -la $R/gpurun_out/r02p | head -30
