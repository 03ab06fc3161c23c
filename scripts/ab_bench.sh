#!/bin/bash
# usage: scripts/ab_bench.sh "<libA> <libB>" <rounds> [flags...]  -- interleaved bench.py runs (the mining workload) on ONE box
libs=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do for l in $libs; do for f in ${@:-0}; do
  out=$(ABC_LIB=$GRAFT_REPO_ROOT/variants/$l ABC_DEBUG_FLAGS=$f timeout -k 10 300 python3 bench.py --steps 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['achieved'],1))") || exit 1
  echo "$l flags=$f $out"
done; done; done
