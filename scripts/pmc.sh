#!/bin/bash
# usage: scripts/pmc.sh <tag> <quick_bench args...>   (run on the GPU box; one rocprofv3 pass per counter set)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 scripts/quick_bench.py "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pass $i done"
done
