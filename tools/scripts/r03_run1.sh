#!/bin/bash
# round 3, GPU call 1: the whole -m gpu suite + the default bench line
O=gpurun_out/r03_1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_gpu_gemm_tiles.py > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -30 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 3000 $O/bench.json
