#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_pmc_conv; mkdir -p $O
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
G2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC"
G3="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS"
G4="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA"
for cfg in "8 0" "8 1" "8 2" "16 0" "17 0"; do
  set -- $cfg; T=$1; A=$2
  dirs=""
  n=0
  for G in "$G1" "$G2" "$G3" "$G4"; do
    n=$((n+1))
    TILE=$T ABLATE=$A rocprofv3 --pmc $G --kernel-trace --output-format csv -d /tmp/pc_${T}_${A}_$n -- python $R/tools/conv3x3_pmc.py 3 > /tmp/pc.log 2>&1
    dirs="$dirs /tmp/pc_${T}_${A}_$n"
  done
  python $R/tools/pmc_collect.py $O/tile${T}_ablate${A}.json $dirs
done
ls $O
