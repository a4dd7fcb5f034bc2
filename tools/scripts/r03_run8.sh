#!/bin/bash
O=gpurun_out/r03_8; mkdir -p $O
T=tests/test_gpu_train_step.py::test_fpn_training_step_gradients_match_autograd
for mode in "A" "RELNET_GEOM_BWD_OLD=1" "RELNET_GEOM_RECOMPUTE_LIBM=1" "RELNET_GEOM_BWD_OLD=1 RELNET_GEOM_RECOMPUTE_LIBM=1"; do
  echo "=== $mode"; env $mode timeout 300 python -m pytest $T -q --tb=short 2>&1 | grep -E "nms_pair_pos|passed|failed" | head -5
done
