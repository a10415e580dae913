#!/bin/bash
# Memory-pipeline PMC passes for the traversal kernels (runs on the GPU box). usage: tools/pmc_mem.sh <tag>
set -u
export TMPDIR=/tmp
TAG=${1:-m}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="bench.py --steps 4 --warmup 1 --no-cpu-baseline --pipes 1"
cd ${GRAFT_REPO_ROOT:-$PWD}
run() { # name counters...
  local n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$n -o $n -- python3 $ARGS > /dev/null 2> $OUT/$n.log || { echo "pass $n failed"; tail -5 $OUT/$n.log; exit 1; }
}
run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE
run stl TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
run tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_GATE_EN1_sum
python3 tools/pmc_quick.py $OUT
