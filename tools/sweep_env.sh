#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   runs the short bench with VAR=v
VAR=$1; shift
for v in "$@"; do
  env $VAR="$v" timeout -k 10 200 python bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('$VAR=$v', 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], 'extend', k['extend']['ms_per_launch'], 'shadow', k['shadow']['ms_per_launch'], 'shade', k['shade']['ms_per_launch'], flush=True)" || exit 1
done
