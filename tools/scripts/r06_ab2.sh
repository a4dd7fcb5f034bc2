# round 6 same-box A/B: weight-gradient products on a side stream (RELNET_WGRAD_OVERLAP = units per flush, 0 = off) at 1 / 2 / 8 images per GPU
O=gpurun_out/r06_ab2; mkdir -p $O
line() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{\"metric')][-1]); print(sys.argv[2], round(d['value'],1), 'img/s', round(d['ms_per_step'],4), 'ms')" "$1" "$2" 2>/dev/null || echo "$2 FAILED"; }
for b in 1 2 8; do
  T="--train --learn-nms --batch $b --steps 40 --warmup 5"
  for k in 0 1 2 4 8 0; do
    RELNET_WGRAD_OVERLAP=$k python bench.py $T > $O/tr_b${b}_k$k.json 2>/dev/null; line $O/tr_b${b}_k$k.json "train_b${b}_wgrad_overlap=$k"
  done
done
