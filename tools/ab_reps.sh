#!/bin/bash
# usage: REPS=4 BENCH_ARGS="..." tools/ab_reps.sh <tag> lib1.so lib2.so ... -- the bench line per library, REPS times, interleaved; the
# last lines give each library's median. Log: gpurun_out/r03/<tag>.log
TAG=$1; shift
mkdir -p gpurun_out/r03
LOG=gpurun_out/r03/$TAG.log
: > $LOG
for rep in $(seq ${REPS:-4}); do
for l in "$@"; do
  PRT_LIB="$PWD/$l" timeout -k 10 300 python bench.py --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('$l', d['value'], k['extend']['ms_per_launch'], k['shadow_extend']['ms_per_launch'], k['shadow']['ms_per_launch'], flush=True)" >> $LOG || exit 1
done
done
python - $LOG <<'PY' >> $LOG
import sys, statistics as st, collections
rows = collections.defaultdict(list)
for line in open(sys.argv[1]):
    f = line.split()
    rows[f[0]].append([float(x) for x in f[1:5]])
for lib, r in rows.items():
    print('median', lib, 'Mrays/s %.0f  extend %.3f  shadow_extend %.3f  shadow %.3f' % tuple(st.median(c) for c in zip(*r)))
PY
cat $LOG
