#!/bin/bash
# usage: tools/sweep_corun.sh -- resident blocks per CU of the traversal kernels x of k_shade x pipes: can the two
# kernel families of different passes share the CUs instead of alternating?
for tb in 5 4 3; do
 for sb in 0 2 1; do
  for pipes in 2 3; do
   PRT_GRID_BLOCKS_PER_CU=$tb PRT_SHADE_BLOCKS_PER_CU=$sb PRT_PIPES=$pipes timeout -k 10 200 python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('trace blocks/CU', $tb, 'shade blocks/CU', $sb, 'pipes', $pipes, 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], flush=True)" || exit 1
  done
 done
done
