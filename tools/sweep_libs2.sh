#!/bin/bash
# usage: tools/sweep_libs2.sh lib1.so ... -- A/B of measurement builds, one iteration stream in flight, per-kernel times
for l in "$@"; do
  PRT_LIB="$PWD/$l" timeout -k 10 200 python bench.py --steps 16 --warmup 8 --no-cpu-baseline --pipes 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('$l', 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], {n: k[n]['ms_per_launch'] for n in ('extend','shadow_extend','shade')}, flush=True)" || exit 1
done
