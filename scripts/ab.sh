#!/bin/bash
# usage: scripts/ab.sh "<libA> <libB> ..." <rounds> <quick_bench args...>  -- interleaved A/B timing on ONE box
libs=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  for l in $libs; do
    for f in ${ABFLAGS:-0}; do
      out=$(ABC_LIB=$GRAFT_REPO_ROOT/variants/$l ABC_DEBUG_FLAGS=$f timeout -k 10 120 python -u scripts/quick_bench.py "$@" 2>/dev/null | tail -1) || exit 1
      echo "$l flags=$f $out"
    done
  done
done
