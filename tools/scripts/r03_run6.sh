#!/bin/bash
O=gpurun_out/r03_6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_step.py tests/test_gpu_relation_bwd.py tests/test_gpu_targets.py tests/test_gpu_ffi_twins.py tests/test_gpu_relation.py tests/test_gpu_mx_facade.py -q --tb=short > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -25
timeout 300 python bench.py --train --learn-nms --steps 10 --warmup 3 > $O/train.json 2> $O/train.err; echo "train rc $?"; python -c "
import json;d=json.loads([l for l in open('$O/train.json') if l.startswith('{')][0]);print('TRAIN', d['value'], d['ms_per_step'])"; tail -3 $O/train.err
timeout 300 python bench.py --train --learn-nms --batch 16 --steps 6 --warmup 2 > $O/train16.json 2> $O/train16.err; python -c "
import json;d=json.loads([l for l in open('$O/train16.json') if l.startswith('{')][0]);print('TRAIN16', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --train --learn-nms --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "prof rc $?"
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/train_kernel_stats.csv; rm -rf $O/prof
