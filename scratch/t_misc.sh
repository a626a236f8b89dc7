set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -u -m pytest tests/test_zz_no_alloc.py tests/test_gpu_remd_device.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -5 &&
timeout -k 10 200 python bench.py --no-cpu-baseline --workload 2xov-remd --steps 40 --warmup 5 | python scratch/show_remd.py 2xov-remd
