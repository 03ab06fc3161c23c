cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for qb in 32 64 16; do
 out=$(ABC_PLAN_QB=$qb timeout -k 10 120 python -u scripts/quick_bench.py 16384 412500 4096 64 bf16 6 2>/dev/null | tail -1)
 echo "plan_qb=$qb $out"
done; done
