#!/bin/bash
O=gpurun_out/r04_10; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_postprocess_topk.py tests/test_gpu_deform.py tests/test_gpu_pipeline.py tests/test_gpu_ffi_twins.py tests/test_gpu_fpn.py -x -q > $O/tests.log 2>&1; tail -6 $O/tests.log
F="--no-cpu-baseline --no-train-line --no-other-configs"
python bench.py $F > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'P'
import json
r=json.loads(open('gpurun_out/r04_10/bench.json').read().strip().splitlines()[-1])
print('value', r['value'], r['ms_per_step']); print('sweep', {k:(v['ms_per_step'] if isinstance(v,dict) else v) for k,v in r.get('batch_sweep').items() if k!='note'})
print(r['kernels_ms'])
print('parity', {k:v for k,v in r['parity'].items() if k in ('proposal_rows_identical','roi_pool_mismatches','detections_matched','detections_gpu','detections_oracle')})
P
python bench.py --dcn --batch 27 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dcn b27', round(d['value'],1), round(d['ms_per_step'],3))"
