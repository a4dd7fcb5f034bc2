#!/bin/bash
mkdir -p gpurun_out/r03_31
timeout 600 python -m torch.distributed.run --nnodes 1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/dist_check.py 2> gpurun_out/r03_31/dist_check.err | grep DIST_CHECK; tail -3 gpurun_out/r03_31/dist_check.err | cut -c1-200
export RELNET_BENCH_ONE_DEVICE=1
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 16 --no-cpu-baseline > gpurun_out/r03_31/two_ranks.json 2> gpurun_out/r03_31/two_ranks.err; echo "rc $?"
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r03_31/two_ranks.json') if l.startswith('{"metric')][-1])
    print('n_gpus', d['n_gpus'], 'value', round(d['value'],1), 'ranks_seen', d['config'].get('ranks_seen_by_rccl'), 'train', d.get('train',{}).get('value'), d.get('train',{}).get('config',{}).get('parallelism'), d.get('train',{}).get('losses'))
except Exception as e: print('FAILED', e)
PY
grep -a "Error\|error" gpurun_out/r03_31/two_ranks.err | head -5
