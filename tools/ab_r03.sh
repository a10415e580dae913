#!/bin/bash
# usage: tools/ab_r03.sh <tag> lib1.so lib2.so ... -- round-3 A/B: quick parity per measurement build (trace vs oracle + brute
# force, image parity on the small scenes), then the default bench line per library, twice, interleaved.
# Log: gpurun_out/r03/<tag>.log
TAG=$1; shift
mkdir -p gpurun_out/r03
LOG=gpurun_out/r03/$TAG.log
{
for l in "$@"; do
  echo "== parity $l"
  PRT_LIB="$PWD/$l" timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_refit.py tests/test_gpu_instances.py -x -q --timeout 300 --timeout-method thread -k "not full_size" 2>&1 | tail -n 3 || exit 1
done
for rep in 1 2; do
for l in "$@"; do
  PRT_LIB="$PWD/$l" timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('$l', 'Mrays/s', d['value'], {n: k[n]['ms_per_launch'] for n in ('extend','shadow_extend','shadow','shade')}, flush=True)" || exit 1
done
done
} > $LOG 2>&1
cat $LOG
