#!/bin/bash
O=gpurun_out/r04_4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_refcuda.py -x -q > $O/refcuda.log 2>&1; tail -5 $O/refcuda.log
F="--no-cpu-baseline --no-train-line --no-other-configs --no-parity --no-batch-sweep"
one() { python - "$1" <<'P'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(r['value'],1), round(r['ms_per_step'],3))
except Exception as e: print(sys.argv[1],'ERR',e)
P
}
export RELNET_DEBUG_KNOBS=1
for i in 1 2; do
  RELNET_GEMM_ASM=0 RELNET_INPLACE_EXPAND=0 python bench.py $F > $O/base_$i.json 2>$O/err.txt; one $O/base_$i.json
  RELNET_GEMM_ASM=1 RELNET_INPLACE_EXPAND=0 python bench.py $F > $O/asm_$i.json 2>$O/err.txt; one $O/asm_$i.json
  RELNET_GEMM_ASM=1 RELNET_INPLACE_EXPAND=1 python bench.py $F > $O/asm_inplace_$i.json 2>$O/err.txt; one $O/asm_inplace_$i.json
done
unset RELNET_DEBUG_KNOBS
( time python bench.py ) > $O/default_full.json 2> $O/default_full.err; tail -5 $O/default_full.err
python - <<'P'
import json
r=json.loads(open('gpurun_out/r04_4/default_full.json').read().strip().splitlines()[-1])
print('value', r['value'], r['ms_per_step']); print('sweep', r.get('batch_sweep')); print('train', {k:v for k,v in r.get('train',{}).items() if k in ('value','ms_per_step','at_16_images_per_gpu')})
print('other', json.dumps(r.get('other_configs'), indent=0)[:3000])
print('parity', {k:v for k,v in r['parity'].items() if k in ('proposal_rows_identical','roi_pool_mismatches','detections_matched','cls_score_max_rel_err')})
P
