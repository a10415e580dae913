#!/bin/bash
# usage: tools/iter.sh <tag> [lib.so ...] -- one development iteration on the GPU box: quick parity, wave statistics and
# bench lines for the in-tree library (and any measurement builds named), log under gpurun_out/r02/<tag>.log
TAG=${1:-it}; shift
mkdir -p gpurun_out/r02
LOG=gpurun_out/r02/$TAG.log
{
  timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_refit.py -x -q --timeout 200 --timeout-method thread -k "not full_size" 2>&1 | tail -n 4 || exit 1
  for l in pathtracer_amd/libprt_hip.so "$@"; do
    echo "== $l"
    PRT_LIB="$PWD/$l" timeout -k 10 120 python tools/simd_util.py 1920 1080 5 2>/dev/null || exit 1
    PRT_LIB="$PWD/$l" timeout -k 10 200 python bench.py --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print('bench', 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], {n: k[n]['ms_per_launch'] for n in k}, flush=True)" || exit 1
  done
} > $LOG 2>&1
cat $LOG
