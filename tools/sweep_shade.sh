#!/bin/bash
# usage: tools/sweep_shade.sh -- resident blocks per CU of k_shade
for nb in 0 4 5 6 7 8 10 12; do
  PRT_SHADE_BLOCKS_PER_CU=$nb timeout -k 10 200 python bench.py --steps 16 --warmup 8 --no-cpu-baseline --pipes 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('shade blocks/CU', $nb, 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], {n: k[n]['ms_per_launch'] for n in ('extend','shadow_extend','shade')}, flush=True)" || exit 1
done
