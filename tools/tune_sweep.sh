#!/bin/bash
# one-at-a-time sweep of the traversal knobs (PRT_TUNE=refillMin,stallMax,leafMin) on the bench workload (default bench:
# 64 iterations per pass); log: gpurun_out/r03/tune_sweep.log
mkdir -p gpurun_out/r03
LOG=gpurun_out/r03/tune_sweep.log
: > $LOG
DEF="16,28,16 8,28,16 12,28,16 24,28,16 32,28,16 16,20,16 16,24,16 16,32,16 16,36,16 16,28,8 16,28,12 16,28,24 16,28,32"
for t in ${@:-$DEF}; do
  PRT_TUNE=$t timeout -k 10 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('$t', d['value'], {n: k[n]['ms_per_launch'] for n in ('extend','shadow_extend','shadow')}, flush=True)" >> $LOG || exit 1
done
cat $LOG
