"""Summarise the rocprofv3 databases written by tools/profile.sh into profiles/<tag>_*.{md,json}."""
import json, os, sqlite3, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)

def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0]

out = [f"# rocprofv3 summary ({tag}) — `python3 bench.py --steps 32 --warmup 16 --no-cpu-baseline --pipes 1` on one MI355X (one pass in flight, so that a kernel's trace duration is its own; every path-tracing launch covers 16 iterations = 33.2 M pixels)\n"]
db = sqlite3.connect(os.path.join(G, "prof_trace", "trace_results.db"))
out.append("## `rocprofv3 --kernel-trace --stats` (top kernels; durations in microseconds)\n")
out.append("| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|")
for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 18"):
    out.append(f"| `{short(name)}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |")
# register / LDS / scratch usage
out.append("\n| kernel | VGPR | SGPR | LDS B | scratch B | grid | block |\n|---|---:|---:|---:|---:|---:|---:|")
seen = set()
for r in db.execute("select name,vgpr_count,sgpr_count,lds_size,scratch_size,grid_x,workgroup_x from kernels"):
    s = short(r[0])
    if s in seen or not s.startswith("prt::"): continue
    seen.add(s); out.append(f"| `{s}` | {r[1]} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]} |")

def per_kernel(dbname, counters):
    d = sqlite3.connect(os.path.join(G, dbname))
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
    for name, cname, val, did in d.execute("select kernel_name,counter_name,value,dispatch_id from counters_collection"):
        acc[short(name)][cname] += val; n[short(name)].add(did)
    return acc, {k: len(v) for k, v in n.items()}

traffic = {}
fetch, nf = per_kernel("prof_fetch/fetch_results.db", ["FETCH_SIZE"])
write, nw = per_kernel("prof_write/write_results.db", ["WRITE_SIZE"])
tcc, nt = per_kernel("prof_tcc/tcc_results.db", [])
sq, ns = per_kernel("prof_sq/sq_results.db", [])
out.append("\n## PMC passes (separate runs: FETCH_SIZE | WRITE_SIZE | TCC_HIT/MISS | SQ_*)\n")
out.append("FETCH_SIZE / WRITE_SIZE are reported in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts 128-byte "
           "requests at 64 B for 16-byte-per-lane loads on gfx950, so the `HBM read` column is FETCH_SIZE x 2. The counters sit at the L2's "
           "fabric side, so Infinity-Cache hits are included: this is traffic leaving L2, an upper bound on HBM traffic.\n")
out.append("| kernel | launches | FETCH_SIZE KiB/launch (raw) | read MB/launch (x2) | WRITE_SIZE KiB/launch | L2 hit rate | VALU busy (SQ_ACTIVE_INST_VALU*4/SQ_BUSY_CYCLES... see below) |\n|---|---:|---:|---:|---:|---:|---:|")
for k in ["prt::k_shadow_extend<false>", "prt::k_extend<false>", "prt::k_shadow<false>", "prt::k_shade", "prt::k_generate", "prt::k_resolve"]:
    if k not in fetch: continue
    f = fetch[k]["FETCH_SIZE"] / nf[k]; w = write[k]["WRITE_SIZE"] / nw[k]
    h = tcc[k].get("TCC_HIT_sum", 0.0); m = tcc[k].get("TCC_MISS_sum", 0.0)
    s = sq[k]
    valu = s.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / max(1.0, s.get("SQ_WAVE_CYCLES", 0.0) * 4.0) if False else 0
    util = s.get("SQ_ACTIVE_INST_VALU", 0.0) / max(1.0, s.get("SQ_WAVE_CYCLES", 1.0))
    out.append(f"| `{k}` | {nf[k]} | {f:.0f} | {f*1024*2/1e6:.2f} | {w:.0f} | {h/max(1.0,h+m):.3f} | {util:.3f} |")
    traffic[k.replace("prt::", "").split("<")[0]] = {"hbm_bytes_per_launch": round(f * 1024 * 2 + w * 1024), "fetch_kib_raw": round(f, 1),
                                                       "write_kib": round(w, 1), "l2_hit_rate": round(h / max(1.0, h + m), 4),
                                                       "valu_active_per_wave_cycle": round(util, 4)}
out.append("\nLast column = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (fraction of resident-wave cycles in which the wave is issuing VALU work).\n")
out.append("### raw SQ counters (sum over launches)\n")
for k in ["prt::k_shadow_extend<false>", "prt::k_extend<false>", "prt::k_shadow<false>", "prt::k_shade"]:
    if k in sq: out.append(f"* `{k}`: " + ", ".join(f"{c}={v:.4g}" for c, v in sorted(sq[k].items())))
open(os.path.join(P, f"{tag}_rocprof_summary.md"), "w").write("\n".join(out) + "\n")
bj = os.path.join(G, "prof_trace.json")
if os.path.exists(bj):
    # the launch size the counters belong to (bench.py scales the traffic to its own launch size with it)
    try:
        r = json.load(open(bj))["roofline"]
        k = r["kernel"].split(" ")[0]
        if k in traffic:
            traffic[k]["rays_per_launch"] = r["standalone"]["rays_per_launch"]
            traffic[k]["hbm_bytes_per_ray"] = round(traffic[k]["hbm_bytes_per_launch"] / r["standalone"]["rays_per_launch"], 2)
    except Exception as e:
        print("no launch size:", e)
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
if os.path.exists(bj):
    open(os.path.join(P, f"{tag}_bench_under_rocprof.json"), "w").write(open(bj).read())
print("\n".join(out))
