#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export AWSEM_FORCE_CELL=1
AWSEM_ASYNC_REPORT=0 timeout -k 10 200 python scratch/cell_async_probe.py 2>&1 | tail -2
AWSEM_ASYNC_REPORT=1 timeout -k 10 200 python scratch/cell_async_probe.py 2>&1 | tail -2
