set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_remd_device.py tests/test_gpu_properties.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -4 &&
timeout -k 10 200 python bench.py --no-cpu-baseline --workload 2xov-remd --steps 40 --warmup 5 | python scratch/show_remd.py 2xov-remd &&
AB_ARGS="" bash tools/ab_lib.sh base
