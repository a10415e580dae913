#!/bin/bash
# one-at-a-time sweep of the traversal knobs (PRT_TUNE=refillMin,stallMax,leafMin) on the bench workload
mkdir -p gpurun_out/r02
LOG=gpurun_out/r02/tune_sweep.log
: > $LOG
DEF="16,24,8 8,24,8 12,24,8 24,24,8 32,24,8 16,12,8 16,16,8 16,32,8 16,40,8 16,24,1 16,24,4 16,24,12 16,24,16"
for t in ${@:-$DEF}; do
  PRT_TUNE=$t timeout -k 10 120 python bench.py --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('$t', d['value'], {n: k[n]['ms_per_launch'] for n in ('extend','shadow_extend')}, flush=True)" >> $LOG || exit 1
done
cat $LOG
