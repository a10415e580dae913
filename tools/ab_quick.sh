#!/bin/bash
# usage: tools/ab_quick.sh lib1.so lib2.so ... -- default bench (64 iterations, one pass) per library, twice, interleaved
for rep in 1 2; do
for l in "$@"; do
  PRT_LIB="$PWD/$l" timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('$l', 'Mrays/s', d['value'], {n: k[n]['ms_per_launch'] for n in ('extend','shadow_extend','shade')}, flush=True)" || exit 1
done
done
