#!/bin/bash
# fixed cost per launch of the sweep: kernel time against tiles per split (rocprofv3 kernel stats, one run per size)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/shard_scan
mkdir -p $out
run() {  # tag nc
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$1 -- python3 scripts/quick_bench.py 8192 $2 4096 64 bf16 4 > $out/$1.log 2>&1
  echo "$1 nc=$2 QB_UB=$QB_UB ABC_PLAN_QB=$ABC_PLAN_QB"; grep -h "sweep16" $out/$1/*/*_kernel_stats.csv | cut -d, -f2-7
}
export QB_UB=-2
run ub_103k 103125
run ub_412k 412500
unset QB_UB
export ABC_PLAN_QB=16
run qb16_412k 412500
run qb16_825k 825000
