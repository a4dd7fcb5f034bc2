#!/bin/bash
T=tests/test_gpu_train_step.py::test_fpn_training_step_gradients_match_autograd
for b in cache16 mfma32 libm; do for y in cache recompute; do
  echo "=== bias=$b y=$y"; RELNET_BWD_BIAS=$b RELNET_BWD_Y=$y timeout 300 python -m pytest $T -q --tb=short 2>&1 | grep -E "AssertionError: |passed|failed" | head -3
done; done
