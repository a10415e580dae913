#!/bin/bash
# Fixed cost of a launch: every kernel's own duration (one pass in flight) at 4 .. 64 iterations per pass (PRT_BATCH), and the
# least-squares line a + b*n per kernel. Log: gpurun_out/r03/tail_fit.log
mkdir -p gpurun_out/r03
LOG=gpurun_out/r03/tail_fit.log
: > $LOG
for b in 4 8 16 32 64; do
  PRT_BATCH=$b timeout -k 10 200 python bench.py --no-cpu-baseline --steps $b --warmup $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_standalone']
print($b, d['value'], json.dumps({n: k[n]['ms_per_launch'] for n in k}), flush=True)" >> $LOG || exit 1
done
python - $LOG <<'PY' >> $LOG
import sys, json
import numpy as np
rows = [l.split(None, 2) for l in open(sys.argv[1]) if l[0].isdigit()]
n = np.array([float(r[0]) for r in rows])
ks = json.loads(rows[0][2]).keys()
for k in ks:
    y = np.array([json.loads(r[2])[k] for r in rows])
    b, a = np.polyfit(n, y, 1)
    print('fit %-16s fixed %.3f ms + %.4f ms per iteration  (%.1f %% of a 20-iteration launch)' % (k, a, b, 100 * a / (a + 20 * b)))
PY
cat $LOG
