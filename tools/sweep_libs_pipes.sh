#!/bin/bash
# usage: tools/sweep_libs_pipes.sh lib.so ... -- A/B of measurement builds with the default pipelining (whole-render rate only)
for l in "$@"; do
  PRT_LIB="$PWD/$l" timeout -k 10 200 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$l', 'pipelined Mrays/s', d['value'], 'ms/step', d['ms_per_step'], flush=True)" || exit 1
done
