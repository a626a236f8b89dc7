set -o pipefail
cd $GRAFT_REPO_ROOT
for q in 1 0; do AWSEM_QUERY_REUSE=$q timeout -k 10 200 python bench.py --no-cpu-baseline --workload 2xov-remd --steps 40 --warmup 5 | python scratch/show_remd.py reuse=$q || exit 1; done
