#!/bin/bash
O=gpurun_out/r04_3; mkdir -p $O
python tests/golden/gen_golden_gpu.py $O/ref_cuda.npz > $O/gen_golden_gpu.log 2>&1; tail -3 $O/gen_golden_gpu.log
F="--no-cpu-baseline --no-train-line"
python bench.py $F > $O/bench_t18.json 2> $O/bench_t18.err; tail -c 400 $O/bench_t18.err
RELNET_GEMM_FORCE_TILE_OFF=1 true
python - <<'P'
import json
r=json.loads(open('gpurun_out/r04_3/bench_t18.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r.get('batch_sweep'))
print({k:v for k,v in r['parity'].items() if k not in ('backbone','worst')})
for c in r['conv_roofline']['top']: print(c)
print(r['kernels_ms'])
P
