#!/bin/bash
# usage: tools/sweep_tune.sh "24,16,24" "16,8,16" ...   (PRT_TUNE = refillMin,stallMax,leafMin)
for t in "$@"; do
  PRT_TUNE="$t" timeout -k 10 200 python bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('$t', 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], 'extend', k['extend']['ms_per_launch'], 'shadow', k['shadow']['ms_per_launch'], 'shade', k['shade']['ms_per_launch'], 'n/ray', d['roofline']['nodes_per_ray'], flush=True)" || exit 1
done
