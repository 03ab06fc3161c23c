#!/bin/bash
# usage (on the GPU box): scripts/profile_round.sh <tag>
# 1. bench line, 2. rocprofv3 kernel-trace stats of the same command, 3. PMC passes (one run each) for the sweep kernel
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 400 python3 bench.py > $out/bench_line.json 2> $out/bench.err
echo "bench done"; cat $out/bench_line.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --no-cpu-baseline > $out/bench_profiled_line.json 2> $out/stats.err
echo "stats done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_$i.log 2>&1
  echo "pmc pass $i done"
done
