# round 6 same-box A/B: share rule of the grouped weight-gradient launches (RELNET_WGRAD_TILES: 1 = stream-K shares + atomics, 2 = whole tiles, 0 = auto)
O=gpurun_out/r06_ab3; mkdir -p $O
export RELNET_DEBUG_KNOBS=1
line() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{\"metric')][-1]); print(sys.argv[2], round(d['value'],1), 'img/s', round(d['ms_per_step'],4), 'ms')" "$1" "$2" 2>/dev/null || echo "$2 FAILED"; }
for b in 1 2; do
  for i in 1 2; do
    for k in 1 2 0; do
      RELNET_WGRAD_TILES=$k python bench.py --train --learn-nms --batch $b --steps 40 --warmup 5 > $O/tr_b${b}_t${k}_$i.json 2>/dev/null; line $O/tr_b${b}_t${k}_$i.json "train_b${b}_wgrad_tiles=$k"
    done
  done
done
cd /tmp; export TMPDIR=/tmp
for k in 1 2; do
  rm -rf /tmp/pw$k; RELNET_WGRAD_TILES=$k rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw$k -- python $GRAFT_REPO_ROOT/bench.py --train --learn-nms --batch 1 --steps 10 --warmup 3 > /dev/null 2>&1
  grep -h "wgrad_streamk\|sgd_update" $(find /tmp/pw$k -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | cut -c1-120
done
