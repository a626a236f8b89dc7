#!/bin/bash
# Extra counter passes on the bench workload (clock, transcendental share, scalar and branch issue).  Usage: bash tools/pmc_extra.sh <tag> [bench args]
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-schedule-legs --steps 3 --warmup 1"
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc_x1_$tag -- $B "$@" > $O/pmc_x1_$tag.log 2>&1 || { tail -5 $O/pmc_x1_$tag.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_TRANS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_SENDMSG SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc_x2_$tag -- $B "$@" > $O/pmc_x2_$tag.log 2>&1 || { tail -5 $O/pmc_x2_$tag.log; exit 2; }
python3 $R/tools/pmc_tail.py pmc $O/pmc_x1_$tag awsem_step_kernel 3 1 | tee $O/${tag}_pmc_extra.txt
python3 $R/tools/pmc_tail.py pmc $O/pmc_x2_$tag awsem_step_kernel 3 51200 | tee -a $O/${tag}_pmc_extra.txt
[ -d $O/prof_$tag ] && python3 $R/tools/pmc_tail.py trace $O/prof_$tag awsem_step_kernel 20
