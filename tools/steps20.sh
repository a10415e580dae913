#!/bin/bash
# the driver's bench form (--steps 20 --warmup 5): iterations per pass x passes in flight
for e in "PRT_BATCH=64" "PRT_BATCH=10" "PRT_BATCH=7" "PRT_BATCH=5" "PRT_BATCH=10 PRT_PIPES=2" "PRT_BATCH=5 PRT_PIPES=4"; do
  for rep in 1 2; do
  env $e timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'], d['config']['iterations_in_flight'], d['config']['iterations_per_pass'], flush=True)" || exit 1
  done
done
