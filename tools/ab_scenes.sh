#!/bin/bash
# usage: tools/ab_scenes.sh <tag> lib1.so lib2.so ... -- configs[3] and configs[4] per library, twice, interleaved.
# Log: gpurun_out/r03/<tag>.log
TAG=$1; shift
mkdir -p gpurun_out/r03
LOG=gpurun_out/r03/$TAG.log
: > $LOG
for rep in 1 2; do
for args in "--scene sponza --bounces 8 --rr" "--scene sanmiguel --width 3840 --height 2160 --steps 16 --warmup 16"; do
for l in "$@"; do
  PRT_LIB="$PWD/$l" timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('$l', '$args', 'Mrays/s', d['value'], {n: k[n]['ms_per_launch'] for n in ('extend','shadow_extend','shadow','shade')}, flush=True)" >> $LOG || exit 1
done
done
done
cat $LOG
