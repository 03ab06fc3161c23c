#!/bin/bash
# usage (on the GPU box): scripts/full_vs_shard.sh [rounds] -- bench.py on the full corpus and on the 412 500-row shard, interleaved on ONE box
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${1:-2}); do
  for nc in 3300000 412500; do
    steps=5; [ $nc = 412500 ] && steps=20
    out=$(timeout -k 10 300 python3 bench.py --nc $nc --steps $steps --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1) || exit 1
    echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('nc=%d  %.1f q/s  sweep %.1f TFLOP/s  frac %.4f' % ($nc, d['value'], d['roofline']['achieved'], d['roofline']['frac']))"
  done
done
