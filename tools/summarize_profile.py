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

out = [f"# rocprofv3 summary ({tag}) — `python3 bench.py --steps 64 --warmup 64 --no-cpu-baseline --pipes 1` on one MI355X (one pass in flight, so that a kernel's trace duration is its own; a path-tracing launch covers up to 64 iterations = 132.7 M pixels)\n"]
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
try:
    tcp, ntcp = per_kernel("prof_tcp/tcp_results.db", [])
except Exception:
    tcp, ntcp = {}, {}
# launch duration and rays per launch of every kernel (kernel trace; rays from the bench line of the traced run)
avg_us = {short(n): a for n, a in db.execute("select name,average from top_kernels")}
# Bytes of coalesced (streamed) 16-byte-per-lane reads per ray of the traversal kernels: queue slot 4 + ray record 32
# (+ 16 for the radiance term a visible shadow ray copies): FETCH_SIZE counts these 128-byte requests at 64 B
# (MI355X_MICROARCH.md), while the divergent 64-byte gathers of nodes and triangles are counted exactly
# (tools/calib_fetch.sh: 65.7 B per 64-byte record, 1.05 TCP->L2 requests). So traffic = FETCH_SIZE + WRITE_SIZE +
# half the streamed reads.
STREAMED_READ_B_PER_RAY = 40.0
L1_MISS_CEILING = 0.41  # lines per clock per CU a CU can pull from L2 with divergent gathers (tools/ubench_gather.hip)
out.append("\n## PMC passes (separate runs: FETCH_SIZE | WRITE_SIZE | TCC_HIT/MISS | SQ_*)\n")
out.append("FETCH_SIZE / WRITE_SIZE are reported in KiB. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts the 128-byte requests of "
           "wide coalesced loads at 64 B; calibrated on this access pattern (tools/calib_fetch.sh, gpurun_out/r02/calib_fetch.log) it counts "
           "divergent 64-byte gathers exactly (65.7 B per 64-byte record from a 2 GB table, 1.05 TCP->L2 requests per record). The "
           "`read MB/launch (x2)` column is therefore an upper bound; `pmc_traffic.json` uses FETCH_SIZE + WRITE_SIZE + half the streamed "
           "reads of the traversal kernels (40 B per ray). The counters sit at the L2's fabric side, so Infinity-Cache hits are included: "
           "traffic leaving L2, an upper bound on HBM traffic.\n")
out.append("| kernel | launches | FETCH_SIZE KiB/launch (raw) | read MB/launch (x2) | WRITE_SIZE KiB/launch | L2 hit rate | VALU busy (SQ_ACTIVE_INST_VALU*4/SQ_BUSY_CYCLES... see below) |\n|---|---:|---:|---:|---:|---:|---:|")
def find(prefix):
    """the non-instrumented single-level instantiation of a kernel, whatever its template argument list looks like"""
    c = [k for k in fetch if k == prefix or (k.startswith(prefix + "<false") and "true" not in k)]
    return c[0] if c else None
WANT = [find(p) for p in ("prt::k_shadow_extend", "prt::k_extend", "prt::k_shadow", "prt::k_shade", "prt::k_generate", "prt::k_resolve")]
for k in WANT:
    if k is None or k not in fetch: continue
    f = fetch[k]["FETCH_SIZE"] / nf[k]; w = write[k]["WRITE_SIZE"] / nw[k]
    h = tcc[k].get("TCC_HIT_sum", 0.0); m = tcc[k].get("TCC_MISS_sum", 0.0)
    s = sq[k]
    valu = s.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / max(1.0, s.get("SQ_WAVE_CYCLES", 0.0) * 4.0) if False else 0
    util = s.get("SQ_ACTIVE_INST_VALU", 0.0) / max(1.0, s.get("SQ_WAVE_CYCLES", 1.0))
    out.append(f"| `{k}` | {nf[k]} | {f:.0f} | {f*1024*2/1e6:.2f} | {w:.0f} | {h/max(1.0,h+m):.3f} | {util:.3f} |")
    e = {"fetch_kib_raw": round(f, 1), "write_kib": round(w, 1), "l2_hit_rate": round(h / max(1.0, h + m), 4),
         "valu_active_per_wave_cycle": round(util, 4),
         "wait_any_per_wave_cycle": round(s.get("SQ_WAIT_ANY", 0.0) / max(1.0, s.get("SQ_WAVE_CYCLES", 1.0)), 4),
         "trace_avg_us": round(avg_us.get(k, 0.0), 2)}
    # SIMD busy: a wave64 VALU instruction holds a SIMD-32 for 2 cycles; 1024 SIMDs; clock from GRBM_GUI_ACTIVE (sum of 8 XCDs)
    if avg_us.get(k) and s.get("SQ_INSTS_VALU") and s.get("GRBM_GUI_ACTIVE") and ns.get(k):
        cyc = s["GRBM_GUI_ACTIVE"] / 8.0 / ns[k]  # shader cycles per launch
        e["valu_busy_simd"] = round(s["SQ_INSTS_VALU"] / ns[k] * 2.0 / (cyc * 1024.0), 4)
        e["clock_GHz"] = round(cyc / (avg_us[k] * 1e3), 3)
    if k in tcp and ntcp.get(k):
        t = tcp[k]
        req = t.get("TCP_TCC_READ_REQ_sum", 0.0) / ntcp[k]
        e["tcp_tcc_read_req_per_launch"] = round(req)
        e["tcp_tcc_read_latency_cycles"] = round(t.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / max(1.0, t.get("TCP_TCC_READ_REQ_sum", 1.0)), 1)
        if t.get("GRBM_GUI_ACTIVE"):
            cyc = t["GRBM_GUI_ACTIVE"] / 8.0 / ntcp[k]
            e["l1_miss_lines_per_clk_per_cu"] = round(req / cyc / 256.0, 4)
            e["l1_miss_ceiling_lines_per_clk_per_cu"] = L1_MISS_CEILING
    traffic[k.replace("prt::", "").split("<")[0]] = e
out.append("\nLast column = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (fraction of resident-wave cycles in which the wave is issuing VALU work).\n")
out.append("### raw SQ counters (sum over launches)\n")
for k in WANT[:4]:
    if k in sq: out.append(f"* `{k}`: " + ", ".join(f"{c}={v:.4g}" for c, v in sorted(sq[k].items())))
open(os.path.join(P, f"{tag}_rocprof_summary.md"), "w").write("\n".join(out) + "\n")
bj = os.path.join(G, "prof_trace.json")
if os.path.exists(bj):
    # the launch size the counters belong to (bench.py scales the traffic to its own launch size with it)
    try:
        r = json.load(open(bj))["roofline"]
        k = r["kernel"].split(" ")[0]
        if k in traffic:
            e, n = traffic[k], r["rays_per_launch"]
            e["rays_per_launch"] = n
            e["fetch_correction"] = "1 (gathers, calibrated) + 20 B per ray for the half-counted streamed reads"
            e["hbm_bytes_per_launch"] = round(e["fetch_kib_raw"] * 1024 + e["write_kib"] * 1024 + 0.5 * STREAMED_READ_B_PER_RAY * n)
            e["hbm_bytes_per_ray"] = round(e["hbm_bytes_per_launch"] / n, 2)
            e["hbm_GB/s_at_trace_duration"] = round(e["hbm_bytes_per_launch"] / (e["trace_avg_us"] * 1e-6) / 1e9, 1)
            e["hbm_frac_of_8TB/s"] = round(e["hbm_GB/s_at_trace_duration"] / 8000.0, 4)
            if "tcp_tcc_read_req_per_launch" in e:
                e["tcp_tcc_read_req_per_ray"] = round(e["tcp_tcc_read_req_per_launch"] / n, 2)
    except Exception as e:
        print("no launch size:", e)
traffic["_tag"] = tag
out.append("\n## Derived per kernel (profiles/pmc_traffic.json)\n")
out.append("```\n" + json.dumps(traffic, indent=1) + "\n```")
open(os.path.join(P, f"{tag}_rocprof_summary.md"), "w").write("\n".join(out) + "\n")
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
if os.path.exists(bj):
    open(os.path.join(P, f"{tag}_bench_under_rocprof.json"), "w").write(open(bj).read())
print("\n".join(out))
