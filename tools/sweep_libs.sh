#!/bin/bash
# usage: tools/sweep_libs.sh lib.so ... -- A/B of measurement builds: rate, per-kernel times, nodes / triangles per ray
for l in "$@"; do
  PRT_LIB="$PWD/$l" timeout -k 10 200 python bench.py --steps 16 --warmup 8 --no-cpu-baseline --pipes 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']; o=d['roofline']['other']
print('$l', 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], {n: k[n]['ms_per_launch'] for n in ('extend','shadow_extend','shade')}, o, 'build', d['config']['bvh_build_ms'], flush=True)" || exit 1
done
