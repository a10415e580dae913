#!/bin/bash
# usage: tools/sweep_occ.sh lib1.so lib2.so ... -- resident blocks per CU x library (measurement builds under gpurun_in/)
for l in "$@"; do
 for nb in 0 5 6 7 8; do
  for pipes in 1 0; do
  PRT_LIB="$PWD/$l" PRT_GRID_BLOCKS_PER_CU=$nb timeout -k 10 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --pipes $pipes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('$l', 'blocks/CU', $nb, 'pipes', $pipes, 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], {n: k[n]['ms_per_launch'] for n in ('extend','shadow_extend','shade')}, flush=True)" || exit 1
  done
 done
done
