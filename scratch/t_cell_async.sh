#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export AWSEM_FORCE_CELL=1
echo "== async on"; timeout -k 10 200 python -u -m pytest tests/test_gpu_drivers.py -x -q --timeout 150 --timeout-method thread -k "test_reports_written" 2>&1 | tail -3
echo "== async off"; AWSEM_ASYNC_REPORT=0 timeout -k 10 200 python -u -m pytest tests/test_gpu_drivers.py -x -q --timeout 150 --timeout-method thread -k "test_reports_written" 2>&1 | tail -3
echo "== async on, no graph"; AWSEM_CELL_GRAPH=0 timeout -k 10 200 python -u -m pytest tests/test_gpu_drivers.py -x -q --timeout 150 --timeout-method thread -k "test_reports_written" 2>&1 | tail -3
