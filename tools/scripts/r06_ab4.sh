#!/bin/bash
# weight gradients of the trunk on a side stream with a PARTIAL persistent grid (RELNET_WGRAD_OVERLAP = units per group, RELNET_WGRAD_SIDE_WGS = workgroups)
for b in ${BATCHES:-1 2}; do
  for cfg in "0 0" "2 32" "2 64" "2 96" "4 32" "4 64" "4 96" "8 64" "0 0"; do
    set -- $cfg
    RELNET_WGRAD_OVERLAP=$1 RELNET_WGRAD_SIDE_WGS=$2 python bench.py --train --learn-nms --batch $b --steps 40 --warmup 5 2>/dev/null | grep -a -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*" | tr "\n" " "; echo " train_b$b overlap=$1 wgs=$2"
  done
done
