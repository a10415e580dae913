#!/bin/bash
# usage: tools/sweep_batch.sh [bench args] -- iterations per pass x pipeline depth
for b in 1 2 4 8 16; do
 for pipes in 1 2 3; do
  PRT_BATCH=$b timeout -k 10 200 python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-profile --pipes $pipes "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('batch', $b, 'pipes', $pipes, 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], flush=True)" || exit 1
 done
done
