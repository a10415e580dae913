#!/bin/bash
# usage: tools/ab_libs.sh lib1.so lib2.so ...  -- A/B of measurement builds under gpurun_in/
# three lines per library: 1080p pipelined, 1080p one iteration in flight, 4K one iteration in flight
one() {
  PRT_LIB="$PWD/$1" timeout -k 10 200 python bench.py --steps 12 --warmup 2 --no-cpu-baseline "${@:3}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('$1', '$2', 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], {n: k[n]['ms_per_launch'] for n in ('extend','shadow','shade')}, 'standalone', (d['roofline'].get('standalone') or {}).get('ms_per_launch'), flush=True)" || exit 1
}
for l in "$@"; do
  one $l pipelined || exit 1
  one $l pipes1 --pipes 1 || exit 1
  one $l 4k-pipes1 --pipes 1 --width 3840 --height 2160 || exit 1
done
