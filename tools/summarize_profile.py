"""Summarise the rocprofv3 databases written by `tools/profile.sh <name> ...` (gpurun_out/prof_<name>/) into
profiles/<tag>_<name>_rocprof_summary.md and the entry `scenes[<name>]` of profiles/pmc_traffic.json.

  python tools/summarize_profile.py <tag> <name>

pmc_traffic.json is stamped with the hash of the kernel sources (pathtracer_amd/buildinfo.py) and every entry with the
workload of the profiled bench line; bench.py uses an entry only when both match its own run."""
import json
import os
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pathtracer_amd.buildinfo import kernel_source_hash  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
name = sys.argv[2] if len(sys.argv) > 2 else "conference"
G = os.path.join(ROOT, "gpurun_out", f"prof_{name}")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
command = open(os.path.join(G, "command.txt")).read().strip() if os.path.exists(os.path.join(G, "command.txt")) else "?"
# the kernel sources the counters were taken on (written by tools/profile.sh on the GPU box)
run_hash = open(os.path.join(G, "kernel_source_hash.txt")).read().strip() if os.path.exists(os.path.join(G, "kernel_source_hash.txt")) else None
if run_hash != kernel_source_hash():
    print(f"WARNING: gpurun_out/prof_{name} was measured on kernel sources {run_hash}, this tree is {kernel_source_hash()}: "
          f"the entry is stamped {run_hash} and bench.py will not use it")
bench = None
try:
    bench = json.loads([l for l in open(os.path.join(G, "trace.json")) if l.startswith("{")][-1])
except Exception as e:  # noqa: BLE001
    print("no bench line of the traced run:", e)


def short(n):
    return n.replace("void ", "").split("(")[0]


out = [f"# rocprofv3 summary ({tag}, {name}) — `python3 {command}` on one MI355X\n",
       "One pass in flight (`--pipes 1`), so that a kernel's trace duration is its own. "
       + (f"Workload: {bench['config']['workload']}; {bench['config']['iterations_per_pass']} iterations per pass; "
          f"bench line of the traced run: {bench['value']} Mrays/s.\n" if bench else "\n")]
db = sqlite3.connect(os.path.join(G, "trace", "trace_results.db"))
out.append("## `rocprofv3 --kernel-trace --stats` (top kernels; durations in microseconds)\n")
out.append("| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|")
for kname, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 18"):
    out.append(f"| `{short(kname)}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |")
out.append("\nLaunch geometry as the profiler saw it:\n")
out.append("| kernel | arch VGPR | accum VGPR | SGPR | LDS B | scratch B | grid | block |\n|---|---:|---:|---:|---:|---:|---:|---:|")
seen = set()
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
acc_col = "accum_vgpr_count" if "accum_vgpr_count" in cols else "0"
arch_col = "arch_vgpr_count" if "arch_vgpr_count" in cols else "vgpr_count"
for r in db.execute(f"select name,{arch_col},{acc_col},sgpr_count,lds_size,scratch_size,grid_x,workgroup_x from kernels"):
    s = short(r[0])
    if s in seen or not s.startswith("prt::"):
        continue
    seen.add(s)
    out.append(f"| `{s}` | {r[1]} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]} |")
try:
    import kernel_resources

    out.append("\nRegisters and occupancy as the compiler reports them for these sources (`tools/kernel_resources.py`; waves per SIMD = "
               "what the registers allow, the LDS column x resident blocks must also fit the CU's 160 KiB):\n")
    out.append(kernel_resources.table(kernel_resources.resources()))
except Exception as e:  # noqa: BLE001
    out.append(f"\n(no compiler resource table: {e})")


def per_kernel(sub):
    d = sqlite3.connect(os.path.join(G, sub, f"{sub}_results.db"))
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    for kname, cname, val, did in d.execute("select kernel_name,counter_name,value,dispatch_id from counters_collection"):
        acc[short(kname)][cname] += val
        n[short(kname)].add(did)
    return acc, {k: len(v) for k, v in n.items()}


fetch, nf = per_kernel("fetch")
write, nw = per_kernel("write")
tcc, nt = per_kernel("tcc")
sq, ns = per_kernel("sq")
try:
    tcp, ntcp = per_kernel("tcp")
except Exception:  # noqa: BLE001
    tcp, ntcp = {}, {}
avg_us = {short(n): a for n, a in db.execute("select name,average from top_kernels")}
# Bytes of coalesced (streamed) 16-byte-per-lane reads per ray of the traversal kernels: queue slot 4 + ray record 32
# (+ 16 for the radiance term a visible shadow ray copies): FETCH_SIZE counts these 128-byte requests at 64 B
# (MI355X_MICROARCH.md), while the divergent 64-byte gathers of nodes and triangles are counted exactly
# (tools/calib_fetch.sh: 65.7 B per 64-byte record, 1.05 TCP->L2 requests). So traffic = FETCH_SIZE + WRITE_SIZE +
# half the streamed reads.
STREAMED_READ_B_PER_RAY = 40.0
L1_MISS_CEILING = 0.41  # lines per clock per CU a CU can pull from L2 with divergent gathers (tools/ubench_gather.hip)
out.append("\n## PMC passes (separate runs: FETCH_SIZE | WRITE_SIZE | TCC_HIT/MISS | SQ_* | TCP_*)\n")
out.append("FETCH_SIZE / WRITE_SIZE are reported in KiB. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts the 128-byte requests of "
           "wide coalesced loads at 64 B; calibrated on this access pattern (tools/calib_fetch.sh, profiles/r02_calib_fetch.log) it counts "
           "divergent 64-byte gathers exactly (65.7 B per 64-byte record from a 2 GB table, 1.05 TCP->L2 requests per record). The "
           "`read MB/launch (x2)` column is therefore an upper bound; `pmc_traffic.json` uses FETCH_SIZE + WRITE_SIZE + half the streamed "
           "reads of the traversal kernels (40 B per ray). The counters sit at the L2's fabric side, so Infinity-Cache hits are included: "
           "traffic leaving L2, an upper bound on HBM traffic.\n")
out.append("| kernel | launches | FETCH_SIZE KiB/launch (raw) | read MB/launch (x2) | WRITE_SIZE KiB/launch | L2 hit rate | "
           "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES |\n|---|---:|---:|---:|---:|---:|---:|")


def find(prefix):
    """the non-instrumented instantiation of a kernel that ran (single-level unless the scene is two-level)"""
    c = [k for k in fetch if k == prefix or (k.startswith(prefix + "<false"))]
    c.sort(key=lambda k: -nf.get(k, 0))
    return c[0] if c else None


traffic = {}
WANT = [find(p) for p in ("prt::k_shadow_extend", "prt::k_extend", "prt::k_shadow", "prt::k_shade", "prt::k_generate", "prt::k_resolve")]
for k in WANT:
    if k is None or k not in fetch:
        continue
    f = fetch[k]["FETCH_SIZE"] / nf[k]
    w = write[k]["WRITE_SIZE"] / nw[k]
    h = tcc[k].get("TCC_HIT_sum", 0.0)
    m = tcc[k].get("TCC_MISS_sum", 0.0)
    s = sq[k]
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
out.append("\nLast column = fraction of resident-wave cycles in which the wave is issuing VALU work.\n")
out.append("### raw SQ counters (sum over launches)\n")
for k in WANT[:4]:
    if k in sq:
        out.append(f"* `{k}`: " + ", ".join(f"{c}={v:.4g}" for c, v in sorted(sq[k].items())))
if bench and bench.get("roofline"):
    # the launch size the counters belong to (bench.py scales the traffic to its own launch size with it)
    r = bench["roofline"]
    k = r["kernel"].split(" ")[0]
    if k in traffic:
        e, n = traffic[k], r["rays_per_launch"]
        e["rays_per_launch"] = n
        e["fetch_correction"] = "1 (gathers, calibrated) + 20 B per ray for the half-counted streamed reads"
        e["hbm_bytes_per_launch"] = round(e["fetch_kib_raw"] * 1024 + e["write_kib"] * 1024 + 0.5 * STREAMED_READ_B_PER_RAY * n)
        e["hbm_bytes_per_ray"] = round(e["hbm_bytes_per_launch"] / n, 2)
        e["hbm_GB/s_at_trace_duration"] = round(e["hbm_bytes_per_launch"] / (e["trace_avg_us"] * 1e-6) / 1e9, 1)
        e["hbm_frac_of_8TB/s"] = round(e["hbm_GB/s_at_trace_duration"] / 8000.0, 4)
        if "tcp_tcc_read_req_per_ray" not in e and "tcp_tcc_read_req_per_launch" in e:
            e["tcp_tcc_read_req_per_ray"] = round(e["tcp_tcc_read_req_per_launch"] / n, 2)
        # what bounds it: the share of L1 misses that leave L2 and their mean latency say where the gathers are served
        lat = e.get("tcp_tcc_read_latency_cycles")
        e["verdict"] = (
            f"{e['hbm_frac_of_8TB/s']:.2f} of the HBM peak leaves L2 (Infinity-Cache hits included), L2 hit rate {e['l2_hit_rate']:.2f}, "
            f"mean L1->L2 read latency {lat} cycles (L2 hit ~300, Infinity Cache ~545, HBM ~900+ on an idle chip), VALU busy "
            f"{e.get('valu_busy_simd')}, a wave waits {e['wait_any_per_wave_cycle']:.2f} of its life: "
            + ("latency-bound on L2 / Infinity-Cache gathers, not HBM-bound" if e["hbm_frac_of_8TB/s"] < 0.5 else "HBM-bandwidth-bound"))
entry = {"workload": (bench or {}).get("config", {}).get("workload"), "command": command, "tag": tag,
         "kernel_source_hash": run_hash,
         "iterations_per_pass": (bench or {}).get("config", {}).get("iterations_per_pass"), "kernels": traffic}
pj = os.path.join(P, "pmc_traffic.json")
allj = {}
if os.path.exists(pj):
    try:
        allj = json.load(open(pj))
    except Exception:  # noqa: BLE001
        allj = {}
if "scenes" not in allj:
    allj = {"scenes": {}}
allj.pop("kernel_source_hash", None)  # (round 3a files carried one stamp for all scenes)
allj["scenes"][name] = entry
json.dump(allj, open(pj, "w"), indent=1)
out.append(f"\n## Derived per kernel (profiles/pmc_traffic.json, scenes[{name!r}], kernel sources {run_hash})\n")
out.append("```\n" + json.dumps(entry, indent=1) + "\n```")
open(os.path.join(P, f"{tag}_{name}_rocprof_summary.md"), "w").write("\n".join(out) + "\n")
if bench:
    open(os.path.join(P, f"{tag}_{name}_bench_under_rocprof.json"), "w").write(json.dumps(bench) + "\n")
print("\n".join(out))
