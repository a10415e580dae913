#!/bin/bash
# Quick PMC passes for kernel tuning (runs on the GPU box). usage: tools/pmc_quick.sh <tag>
set -u
export TMPDIR=/tmp
TAG=${1:-q}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="bench.py --steps 4 --warmup 1 --no-cpu-baseline"
cd ${GRAFT_REPO_ROOT:-$PWD}
run() { # name counters...
  local n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$n -o $n -- python3 $ARGS > /dev/null 2> $OUT/$n.log || { echo "pass $n failed"; tail -5 $OUT/$n.log; exit 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
python3 tools/pmc_quick.py $OUT
