#!/bin/bash
# usage: tools/sweep_tune.sh "r,s,l" ... -- PRT_TUNE = refillMin,stallMax,leafMin of the persistent traversal waves
for t in "$@"; do
  PRT_TUNE="$t" timeout -k 10 200 python bench.py --steps 16 --warmup 8 --no-cpu-baseline --pipes 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('PRT_TUNE=$t', 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], {n: k[n]['ms_per_launch'] for n in ('extend','shadow_extend','shadow')}, flush=True)" || exit 1
done
