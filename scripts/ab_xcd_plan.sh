set -e
cd $GRAFT_REPO_ROOT
ABFLAGS="0 4096 8192 12288" bash scripts/ab.sh "libbase.so" 2 8192 3300000 4096 64 bf16 4
echo "--- 16384 queries, plan_qb 32 vs 64"
for r in 1 2; do for qb in 32 64; do
 out=$(ABC_LIB=$GRAFT_REPO_ROOT/variants/libbase.so ABC_PLAN_QB=$qb timeout -k 10 120 python -u scripts/quick_bench.py 16384 3300000 4096 64 bf16 3 2>/dev/null | tail -1)
 echo "plan_qb=$qb $out"
done; done
