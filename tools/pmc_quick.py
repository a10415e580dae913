"""Per-kernel, per-launch counter means from the rocprofv3 databases of tools/pmc_quick.sh."""
import glob, os, sqlite3, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(set))
for db in glob.glob(os.path.join(out, "*", "*_results.db")):
    d = sqlite3.connect(db)
    for name, cname, val, did in d.execute("select kernel_name,counter_name,value,dispatch_id from counters_collection"):
        k = name.replace("void ", "").split("(")[0]
        acc[k][cname] += val; n[k][cname].add(did)
for k in sorted(acc):
    if not any(s in k for s in ("k_extend<false>", "k_shadow<false>", "k_shade", "k_ao_shade")): continue
    print(k)
    for c in sorted(acc[k]):
        print(f"   {c:36s} {acc[k][c] / max(1, len(n[k][c])):16.4g} /launch ({len(n[k][c])} launches)")
