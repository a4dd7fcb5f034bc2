cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02p
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $R/bench.py --no-cpu-baseline --no-kernel-timing --no-parity --no-batch-sweep --no-train-line > /tmp/p_bench.log 2>&1
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02p/bench_b54_kernel_stats.csv
cp $(find /tmp/p_bench -name "*domain_stats.csv" | head -1) $R/gpurun_out/r02p/bench_b54_domain_stats.csv 2>/dev/null
tail -1 /tmp/p_bench.log | cut -c1-160
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_train -- python $R/bench.py --train --learn-nms --batch 8 --steps 4 --warmup 1 --no-graph > /tmp/p_train.log 2>&1
cp $(find /tmp/p_train -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02p/train_lnms_b8_kernel_stats.csv
tail -1 /tmp/p_train.log | cut -c1-160
cd $R && python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_final.json
