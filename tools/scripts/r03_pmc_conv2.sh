#!/bin/bash
# HBM / fabric traffic of the res4 3x3 launch: k order 0 vs 1 (tile 8), A-window schedule (tile 17)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_pmc_conv; mkdir -p $O
for cfg in "8 0" "8 1" "17 1"; do
  set -- $cfg; T=$1; K=$2
  dirs=""
  for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $G | cut -d' ' -f1)
    TILE=$T RELNET_GEMM_KORDER=$K rocprofv3 --pmc $G --kernel-trace --output-format csv -d /tmp/pf_${T}_${K}_$tag -- python $R/tools/conv3x3_pmc.py 3 > /tmp/pf.log 2>&1
    dirs="$dirs /tmp/pf_${T}_${K}_$tag"
  done
  python $R/tools/pmc_collect.py $O/traffic_tile${T}_korder${K}.json $dirs
done
ls $O
