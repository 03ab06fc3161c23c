#!/bin/bash
# usage: scripts/ab_diag.sh "<libs>" <flags> <quick_bench args...> -- diagnostics builds: prints the [abc debug] lines too
libs=$1; flags=$2; shift 2
for l in $libs; do
  echo "== $l flags=$flags"
  ABC_LIB=$GRAFT_REPO_ROOT/variants/$l ABC_DEBUG_FLAGS=$flags timeout -k 10 120 python -u scripts/quick_bench.py "$@" 2>&1 | grep -E "abc debug|iter" | tail -5 || exit 1
done
