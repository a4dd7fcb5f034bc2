#!/bin/bash
# fused DCN conv parity + relation backward fix verification + DCN bench
mkdir -p gpurun_out/r03_11
timeout 600 python -m pytest tests/test_gpu_deform.py -x -q --tb=short 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_relation_bwd.py tests/test_gpu_proposal_roi.py -q --tb=short 2>&1 | tail -15
timeout 300 python bench.py --dcn --batch 27 --steps 10 --warmup 3 > gpurun_out/r03_11/dcn27.json 2> gpurun_out/r03_11/dcn27.err; cat gpurun_out/r03_11/dcn27.json
RELNET_DCN_UNFUSED=1 timeout 300 python bench.py --dcn --batch 27 --steps 10 --warmup 3 > gpurun_out/r03_11/dcn27_unfused.json 2>/dev/null; cat gpurun_out/r03_11/dcn27_unfused.json
timeout 300 python bench.py --train --dcn --steps 10 --warmup 3 2>/dev/null | tail -1
