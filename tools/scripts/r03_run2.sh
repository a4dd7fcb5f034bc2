#!/bin/bash
# round 3, GPU call: backward kernels -- targeted tests, training bench, kernel profile of the training step
O=gpurun_out/r03_2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_relation_bwd.py tests/test_gpu_train_step.py tests/test_gpu_targets.py tests/test_gpu_dataset.py tests/test_gpu_fpn.py -q --tb=short -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -40
timeout 300 python bench.py --train --learn-nms --steps 10 --warmup 3 > $O/train.json 2> $O/train.err; echo "train rc $?"; tail -c 600 $O/train.json; tail -3 $O/train.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --train --learn-nms --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "prof rc $?"
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/train_kernel_stats.csv; find $O/prof -name "*.csv" -size +2M -delete; ls -la $O
