#!/bin/bash
# usage (on the GPU box): scripts/profile_round.sh <tag> [part]
#   part 1: bench line, rocprofv3 kernel-trace stats of the same command, PMC passes (one run each) for the sweep kernel,
#           the 412 500-row shard (what one of 8 GPUs sweeps) beside the full corpus
#   part 2: the five BASELINE configs, cfg5 (int8) kernel stats + PMC passes, cfg4 (InfoNCE) kernel stats
set -e
tag=$1
part=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
SQSET="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
if [ "$part" = 1 ]; then
  timeout -k 10 400 python3 bench.py > $out/bench_line.json 2> $out/bench.err
  echo "bench done"; cat $out/bench_line.json
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --no-cpu-baseline > $out/bench_profiled_line.json 2> $out/stats.err
  echo "stats done"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "$SQSET"; do
    i=$((i+1))
    timeout -s KILL 240 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_$i.log 2>&1
    echo "pmc pass $i done"
  done
  timeout -k 10 200 python3 bench.py --nc 412500 --steps 20 --warmup 2 --no-cpu-baseline > $out/shard412k_line.json 2> $out/shard.err
  echo "shard done"; cat $out/shard412k_line.json
  # the vendor GEMM yardstick on this very box (score GEMM alone, bf16 out, no top-k)
  timeout -k 10 200 python3 scripts/gemm_yardstick.py > $out/gemm_yardstick.jsonl 2> $out/yardstick.err
  echo "yardstick done"; cat $out/gemm_yardstick.jsonl
  python3 scripts/pmc_summary.py $out/pmc_1 $out/pmc_2 $out/pmc_3 $out/pmc_4 > $out/pmc_summary.txt 2>/dev/null || true
else
  timeout -k 10 500 python3 scripts/bench_configs.py > $out/configs.jsonl 2> $out/configs.err
  echo "configs done"; cat $out/configs.jsonl
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg5_stats -- python3 scripts/cfg5_bench.py > $out/cfg5_profiled_line.json 2> $out/cfg5_stats.err
  echo "cfg5 stats done"
  i=0
  for set in "FETCH_SIZE" "$SQSET"; do
    i=$((i+1))
    timeout -s KILL 240 rocprofv3 --pmc $set --output-format csv -d $out/cfg5_pmc_$i -- python3 scripts/cfg5_bench.py 10000000 1 > $out/cfg5_pmc_$i.log 2>&1
    echo "cfg5 pmc pass $i done"
  done
  timeout -k 10 200 python3 scripts/nce_bench.py > $out/cfg4_line.txt 2> $out/cfg4.err
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg4_stats -- python3 scripts/nce_bench.py > $out/cfg4_profiled.txt 2> $out/cfg4_stats.err
  echo "cfg4 done"; cat $out/cfg4_line.txt
fi
