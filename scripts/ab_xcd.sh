# A/B of launch shapes on ONE box: ABC_PLAN_QB (query blocks per launch) x ABC_DEBUG_FLAGS bits 12-13 (XCD grouping)
set -e
for r in 1 2; do
for qb in 32 16; do
for f in 0 4096 8192; do
  echo "bf16-head qb=$qb flags=$f $(ABC_PLAN_QB=$qb ABC_DEBUG_FLAGS=$f timeout -k 10 120 python -u scripts/quick_bench.py 16384 3300000 4096 64 bf16 2 2>/dev/null | tail -1)"
done
done
done
