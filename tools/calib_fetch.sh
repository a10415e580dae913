#!/bin/bash
# Calibration of FETCH_SIZE / TCP_TCC_READ_REQ for divergent 16-byte gathers (MI355X_MICROARCH.md, HBM section: "calibrate
# on a known byte count in your own access pattern"): tools/ubench_gather variant A reads 1536 x 256 x 2050 records of
# 64 bytes (four 16-byte loads each) at random from a 2 GB table, so nearly every record leaves L2 AND the Infinity Cache.
set -u
export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/calib
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-$PWD}
for mb in 2000 20; do
  timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch$mb -o f -- tools/ubench_gather $mb A > $OUT/fetch$mb.log 2>&1 || exit 1
  timeout -s KILL 60 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/tcp$mb -o t -- tools/ubench_gather $mb A > $OUT/tcp$mb.log 2>&1 || exit 1
done
python3 - <<'PY'
import glob, os, sqlite3
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "calib")
rec = 1536 * 256 * 2000  # the timed launch (the warm-up launch of 50 rounds is the other dispatch)
for db in sorted(glob.glob(os.path.join(out, "*", "*_results.db"))):
    d = sqlite3.connect(db)
    rows = list(d.execute("select counter_name, value, dispatch_id from counters_collection"))
    by = {}
    for c, v, i in rows:
        by.setdefault(c, {}).setdefault(i, 0.0)
        by[c][i] += v
    for c, per in by.items():
        v = max(per.values())  # the long launch
        print(os.path.basename(os.path.dirname(db)), c, f"{v:.6g}", "per 64-B record:", f"{v / rec:.4f}",
              "(FETCH_SIZE is in KiB: bytes per record = %.1f)" % (v * 1024 / rec) if c == "FETCH_SIZE" else "")
PY
