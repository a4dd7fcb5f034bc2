#!/bin/bash
# round 4, call 1: Infinity-Cache probe + A/B of the res4 sub-batching (same box, same binary)
O=gpurun_out/r04_1; mkdir -p $O
python tools/llc_probe.py > $O/llc.json 2> $O/llc.err
F="--no-cpu-baseline --no-batch-sweep --no-train-line"
RELNET_STAGE_SPLIT=0 python bench.py $F --no-parity > $O/a_nosplit.json 2> $O/a.err
python bench.py $F > $O/b_split_inplace.json 2> $O/b.err
RELNET_STAGE_SPLIT=4:2 RELNET_INPLACE_EXPAND=0 python bench.py $F --no-parity > $O/c_split_outofplace.json 2> $O/c.err
RELNET_STAGE_SPLIT=0 python bench.py $F --no-parity --batch 27 > $O/d_b27.json 2> $O/d.err
RELNET_STAGE_SPLIT=0 python bench.py $F --no-parity > $O/a2_nosplit.json 2> $O/a2.err
python bench.py $F --no-parity > $O/b2_split_inplace.json 2> $O/b2.err
tail -c 300 $O/*.err
for f in $O/[a-d]*.json; do echo $f; python - "$f" <<'P'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(r['value'], r['ms_per_step'], r['config']['images_per_gpu_per_step'], {k:v for k,v in r.get('kernels_ms',{}).items() if 'conv2d' in k or 'chain' in k})
except Exception as e: print('ERR',e)
P
done
cat $O/llc.json
