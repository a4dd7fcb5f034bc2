#!/bin/bash
F="--no-cpu-baseline --no-parity --no-train-line --no-batch-sweep --no-kernel-timing"
OLD=$GRAFT_REPO_ROOT/relation-networks-for-object-detection_amd/librelnet_hip_old.so
for i in 1 2; do
  RELNET_LIB=$OLD python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old gemm.hip', round(d['value'],1), round(d['ms_per_step'],3))"
  RELNET_GEMM_KORDER=0 python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new, korder 0', round(d['value'],1), round(d['ms_per_step'],3))"
  python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new (default)', round(d['value'],1), round(d['ms_per_step'],3))"
done
