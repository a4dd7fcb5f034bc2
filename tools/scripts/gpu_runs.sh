#!/bin/bash
# The gpurun command lines behind the numbers in DESIGN.md / profiles/ (one parametrised script instead of one scratch file per run):
#     gpurun --timeout 2400 -- 'bash tools/scripts/gpu_runs.sh <what> [args]'        outputs under gpurun_out/r04_<what>/  (or TAG=...)
#   suite      whole GPU test suite
#   bench      the driver's default line (python bench.py), full JSON
#   ab KNOB    same-box A/B of the default step with a kernel-selection knob off / on, twice
#              (KNOB = RELNET_GEMM_ASM | RELNET_INPLACE_EXPAND | RELNET_STAGE_SPLIT=4:2 | RELNET_GEMM_KORDER)
#   tiles T,.. parity tests of the forced tiles, then their per-layer times at 54 images (tools/bench_tiles.py), e.g. `tiles 8,18,19`
#   numbers    every other graph's rate (plain 2FC, learn-NMS, DCN, FPN, training variants): one JSON line each
#   prof       rocprofv3 --kernel-trace --stats of the default bench (54 images), the 1-image step and the training step
#   pmc_attn   FETCH_SIZE / WRITE_SIZE / SQ counter passes on the relation-attention kernel (tools/attn_only.py) -> attention_pmc_raw.json
#   golden     tests/golden/gen_golden_gpu.py (reference CUDA kernels compiled for gfx950) -> ref_cuda.npz
#   probes     tools/llc_probe.py + tools/fill_probe.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
WHAT=$1; shift
TAG=${TAG:-r06_$WHAT}
O=$R/gpurun_out/$TAG; mkdir -p $O
F="--no-cpu-baseline --no-train-line --no-other-configs --no-parity --no-batch-sweep --batch 54"
line() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{\"metric')][-1]); print(sys.argv[2], round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config'].get('images_per_gpu_per_step'))" "$1" "$2" 2>/dev/null || echo "$2 FAILED"; }
case $WHAT in
  suite) ( time timeout 1800 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log ;;
  bench) ( time python bench.py ) > $O/bench.json 2> $O/bench.err; tail -4 $O/bench.err; line $O/bench.json default ;;
  ab)
    KNOB=$1; NAME=${KNOB%%=*}; ON=${KNOB#*=}; [ "$ON" = "$KNOB" ] && ON=1
    export RELNET_DEBUG_KNOBS=1
    for i in 1 2; do
      env $NAME=0 python bench.py $F > $O/off_$i.json 2>/dev/null; line $O/off_$i.json "$NAME=0"
      env $NAME=$ON python bench.py $F > $O/on_$i.json 2>/dev/null; line $O/on_$i.json "$NAME=$ON"
    done ;;
  tiles)
    RELNET_TEST_TILES=$1 timeout 900 python -m pytest tests/test_gpu_gemm_tiles.py -x -q > $O/tiles.log 2>&1; tail -3 $O/tiles.log
    TILES=$1,$1 timeout 600 python tools/bench_tiles.py 54 > $O/bench_tiles_b54.txt 2>&1; cat $O/bench_tiles_b54.txt ;;
  numbers)
    G="--no-cpu-baseline --no-parity --no-batch-sweep --no-train-line --no-other-configs --no-kernel-timing"
    run() { n=$1; shift; timeout 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err; line $O/$n.json $n; }
    run plain2fc $G --no-relation; run lnms27 $G --learn-nms --batch 27; run dcn27 $G --dcn --batch 27
    run dcn_lnms27 $G --dcn --learn-nms --batch 27; run fpn8 $G --fpn --batch 8; run fpn_lnms8 $G --fpn --learn-nms --batch 8
    run train8 --train --steps 10 --warmup 3; run train_lnms8 --train --learn-nms --steps 10 --warmup 3
    run train_lnms16 --train --learn-nms --batch 16 --steps 6 --warmup 2; run train_lnms32 --train --learn-nms --batch 32 --steps 5 --warmup 2
    run train_dcn_lnms8 --train --dcn --learn-nms --steps 6 --warmup 2; run train_fpn_lnms2 --train --fpn --learn-nms --batch 2 --steps 6 --warmup 2 ;;
  prof)
    cd /tmp && export TMPDIR=/tmp
    P="--no-cpu-baseline --no-kernel-timing --no-parity --no-batch-sweep --no-train-line --no-other-configs"
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p108 -- python $R/bench.py $P --batch 108 > /tmp/p108.log 2>&1
    cp $(find /tmp/p108 -name "*kernel_stats.csv" | head -1) $O/bench_b108_kernel_stats.csv
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p54 -- python $R/bench.py $P --batch 54 > /tmp/p54.log 2>&1
    cp $(find /tmp/p54 -name "*kernel_stats.csv" | head -1) $O/bench_b54_kernel_stats.csv
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $R/bench.py $P --batch 1 --steps 50 > /tmp/p1.log 2>&1
    cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/bench_b1_kernel_stats.csv
    rm -rf /tmp/pt_prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_prof -- python $R/bench.py --train --learn-nms --batch 8 --steps 10 --warmup 3 > /tmp/pt.log 2>&1
    cp $(find /tmp/pt_prof -name "*kernel_stats.csv" | head -1) $O/train_lnms_b8_kernel_stats.csv; ls -la $O ;;
  pmc_attn)
    cd /tmp && export TMPDIR=/tmp
    for NB in ${ATTN_BATCHES:-108 54}; do
      for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
        tag=$(echo $C | cut -d' ' -f1)
        rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pa${NB}_$tag -- python $R/tools/attn_only.py $NB 6 > /tmp/pa.log 2>&1; tail -1 /tmp/pa.log
      done
      python $R/tools/pmc_collect.py $O/attention_pmc_raw_b$NB.json /tmp/pa${NB}_FETCH_SIZE /tmp/pa${NB}_WRITE_SIZE /tmp/pa${NB}_SQ_VALU_MFMA_BUSY_CYCLES
    done; ls -la $O ;;
  trainsuite)   # the tests of the training step and its building blocks (round 5: chain forward, mask epilogue, GradSink, buckets)
    ( time timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_step.py tests/test_gpu_relation_bwd.py tests/test_gpu_two_ranks.py tests/test_gpu_bottleneck.py -q -m gpu ) > $O/pytest_train.log 2>&1; tail -15 $O/pytest_train.log ;;
  trainab)      # same-box A/B of the configs[2] training step (8 images): every round-5 form switched off one at a time, then all off
    T="--train --learn-nms --steps ${STEPS:-10} --warmup 3 --batch ${BATCH:-8}"
    run() { n=$1; shift; env "$@" timeout 400 python bench.py $T > $O/$n.json 2> $O/$n.err; line $O/$n.json $n; }
    run all_on A=1; run no_chain RELNET_TRAIN_CHAIN=0; run no_mask RELNET_TRAIN_MASK_EPI=0; run no_overlap RELNET_WGRAD_OVERLAP=0
    run overlap1 RELNET_WGRAD_OVERLAP=1; run overlap2 RELNET_WGRAD_OVERLAP=2; run overlap8 RELNET_WGRAD_OVERLAP=8
    run all_off RELNET_TRAIN_CHAIN=0 RELNET_TRAIN_MASK_EPI=0 RELNET_WGRAD_OVERLAP=0 RELNET_REL_SINK=0; run all_on_again A=1 ;;
  proftrain)
    cd /tmp && export TMPDIR=/tmp
    # EXTRA="--dcn" (configs[3]) / EXTRA="--fpn" BATCH=2 (configs[4]) profile the other training graphs: NAME=train_dcn_lnms ...
    PD=/tmp/pt_${NAME:-train_lnms}_b${BATCH:-8}; rm -rf $PD      # (a fresh directory per run: `find | head -1` below must not pick up an earlier trace)
    rocprofv3 --kernel-trace --stats --output-format csv -d $PD -- python $R/bench.py --train --learn-nms $EXTRA --batch ${BATCH:-8} --steps 10 --warmup 3 > /tmp/pt.log 2>&1
    cp $(find $PD -name "*kernel_stats.csv" | head -1) $O/${NAME:-train_lnms}_b${BATCH:-8}_kernel_stats.csv; tail -2 /tmp/pt.log; ls -la $O ;;
  pmc_trunk)    # round 5: counters of the asm ring tile (res4 3x3, res5 3x3) and of chain256_roles_kernel, 54 images, separate passes per group
                # (the training step's shapes: TAG=r05_pmc_trunk8 IMAGES=8 TILE=4 KERNELS="res4_3x3 chain256" ... pmc_trunk, folded with `pmc_trunk_fold.py trunk8`)
    cd /tmp && export TMPDIR=/tmp
    for K in ${KERNELS:-res4_3x3 res5_3x3 chain256}; do
      i=0
      for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
        i=$((i+1))
        timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pk_${K}_$i -- python $R/tools/kernel_pmc.py $K ${IMAGES:-54} 4 > /tmp/pk.log 2>&1 || echo "pass $K/$i ($C) failed: $(tail -2 /tmp/pk.log)"
      done
      python $R/tools/pmc_collect.py $O/${K}_pmc_raw.json /tmp/pk_${K}_*
      # kernel durations of the same launches (from the kernel trace of pass 1)
      python - $O/${K}_pmc_raw.json /tmp/pk_${K}_1 <<'PY'
import csv, glob, json, os, sys
out, d = sys.argv[1], sys.argv[2]
j = json.load(open(out))
dur = {}
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        dur.setdefault(r['Kernel_Name'], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in dur.items():
    if k in j:
        v = sorted(v)
        j[k]['duration_ns'] = {'launches': len(v), 'median': v[len(v) // 2], 'min': v[0], 'max': v[-1]}
json.dump(j, open(out, 'w'), indent=1, sort_keys=True)
PY
    done; ls -la $O ;;
  golden) python tests/golden/gen_golden_gpu.py $O/ref_cuda.npz ;;
  probes) python tools/llc_probe.py > $O/llc.json; python tools/fill_probe.py > $O/fill_probe.txt; cat $O/fill_probe.txt ;;
  *) echo "unknown run '$WHAT'"; exit 2 ;;
esac
