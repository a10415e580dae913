#!/bin/bash
# Runs on the GPU box: kernel-trace stats and PMC passes (one counter group per run) of one bench workload.
#   tools/profile.sh <name> [bench.py arguments]      e.g.  tools/profile.sh sanmiguel --scene sanmiguel --width 3840 --height 2160 --steps 16 --warmup 16
# Output: gpurun_out/prof_<name>/{trace,fetch,write,tcc,sq,tcp}/ + the bench line of the traced run (trace.json).
# One pass in flight (--pipes 1), so that a kernel's trace duration is its own. Summarise with tools/summarize_profile.py.
set -u
export TMPDIR=/tmp
NAME=${1:-conference}; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$NAME
if [ $# -eq 0 ]; then set -- --steps 64 --warmup 64; fi
ARGS="bench.py --no-cpu-baseline --pipes 1 $*"
T=${PROFILE_TIMEOUT:-300}
cd $ROOT
rm -rf $OUT; mkdir -p $OUT
echo "$ARGS" > $OUT/command.txt
python3 -c "from pathtracer_amd.buildinfo import kernel_source_hash as h; print(h())" > $OUT/kernel_source_hash.txt
echo "== kernel trace"; timeout -s KILL $T rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.json 2> $OUT/trace.log || exit 1
echo "== pmc FETCH_SIZE"; timeout -s KILL $T rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 $ARGS > /dev/null 2> $OUT/fetch.log || exit 1
echo "== pmc WRITE_SIZE"; timeout -s KILL $T rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- python3 $ARGS > /dev/null 2> $OUT/write.log || exit 1
echo "== pmc TCC hit/miss"; timeout -s KILL $T rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/tcc -o tcc -- python3 $ARGS > /dev/null 2> $OUT/tcc.log || exit 1
echo "== pmc SQ"; timeout -s KILL $T rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/sq -o sq -- python3 $ARGS > /dev/null 2> $OUT/sq.log || exit 1
echo "== pmc TCP"; timeout -s KILL $T rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE -d $OUT/tcp -o tcp -- python3 $ARGS > /dev/null 2> $OUT/tcp.log || exit 1
ls -la $OUT/*/*.db
