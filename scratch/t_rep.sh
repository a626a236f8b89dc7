set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_gpu_drivers.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -8 &&
timeout -k 10 200 python scratch/e2e_run.py && timeout -k 10 200 python scratch/e2e_run.py && AWSEM_ASYNC_REPORT=0 timeout -k 10 200 python scratch/e2e_run.py
