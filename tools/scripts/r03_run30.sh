#!/bin/bash
# the N > 1 code path on a one-GPU box: 2 ranks on cuda:0, gloo collectives
export RELNET_BENCH_ONE_DEVICE=1
mkdir -p gpurun_out/r03_30
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 16 > gpurun_out/r03_30/two_ranks.json 2> gpurun_out/r03_30/two_ranks.err; echo "rc $?"; tail -3 gpurun_out/r03_30/two_ranks.err | cut -c1-300
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r03_30/two_ranks.json') if l.startswith('{"metric')][-1])
    print('n_gpus', d['n_gpus'], 'value', round(d['value'],1), 'ranks_seen', d['config'].get('ranks_seen_by_rccl'), 'train', d.get('train',{}).get('value'), d.get('train',{}).get('config',{}).get('parallelism'))
except Exception as e: print('FAILED', e)
PY
timeout 600 python bench.py --gpus 2 --train --learn-nms --steps 4 --warmup 2 2> gpurun_out/r03_30/train2.err | tail -1 | cut -c1-400; tail -2 gpurun_out/r03_30/train2.err | cut -c1-300
