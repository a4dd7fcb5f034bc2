# round 6 same-box A/Bs: split-K (RELNET_GEMM_SPLITK: 1 = off, 0 = default = fc_new_1 only, -2 = + rpn_conv_3x3) on the one-image inference step; bash tools/scripts/r06_ab.sh
O=gpurun_out/r06_ab; mkdir -p $O
F="--no-cpu-baseline --no-train-line --no-other-configs --no-parity --no-batch-sweep --no-kernel-timing --batch 1 --steps 2000 --warmup 50"
line() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{\"metric')][-1]); print(sys.argv[2], round(d['value'],1), 'img/s', round(d['ms_per_step'],4), 'ms')" "$1" "$2" 2>/dev/null || echo "$2 FAILED"; }
export RELNET_DEBUG_KNOBS=1
for i in 1 2 3; do
  for k in 1 0 -2; do
    RELNET_GEMM_SPLITK=$k python bench.py $F > $O/inf_${k}_$i.json 2>/dev/null; line $O/inf_${k}_$i.json inf_b1_splitk=$k
  done
done
