#!/bin/bash
# usage: [SIMD_ARGS="W H bounces scene"] tools/simd_util_lib.sh lib.so ... -- wave-level step statistics per measurement build
for l in "$@"; do echo "== $l"; PRT_LIB="$PWD/$l" timeout -k 10 120 python tools/simd_util.py ${SIMD_ARGS:-1920 1080 5} || exit 1; done
