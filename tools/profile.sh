#!/bin/bash
# Runs on the GPU box: kernel-trace stats and PMC passes of the bench workload. Output under gpurun_out/prof_*.
set -u
export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out
ARGS="bench.py --steps 64 --warmup 64 --no-cpu-baseline --pipes 1"
cd ${GRAFT_REPO_ROOT:-$PWD}
echo "== kernel trace"; timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o trace -- python3 $ARGS > $OUT/prof_trace.json 2> $OUT/prof_trace.log || exit 1
echo "== pmc FETCH_SIZE"; timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- python3 $ARGS > /dev/null 2> $OUT/prof_fetch.log || exit 1
echo "== pmc WRITE_SIZE"; timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write -o write -- python3 $ARGS > /dev/null 2> $OUT/prof_write.log || exit 1
echo "== pmc TCC hit/miss"; timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/prof_tcc -o tcc -- python3 $ARGS > /dev/null 2> $OUT/prof_tcc.log || exit 1
echo "== pmc SQ"; timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/prof_sq -o sq -- python3 $ARGS > /dev/null 2> $OUT/prof_sq.log || exit 1
echo "== pmc TCP"; timeout -s KILL 240 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE -d $OUT/prof_tcp -o tcp -- python3 $ARGS > /dev/null 2> $OUT/prof_tcp.log || exit 1
ls $OUT/prof_*/*.db
