"""Rows of BASELINE.md's result table, measured on the GPU box: HIP path and the CPU oracle (bounded samples) per
configuration of BASELINE.json. usage: python tools/baseline_table.py [--big]  (--big adds the 10 M-triangle scene)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from pathtracer_amd import capi, scenes, uniforms
import pyoracle

THREADS = max(1, len(os.sched_getaffinity(0)))
rows = []


def rmse_rel(a, b):
    return float(np.mean((a.astype(np.float64) - b) ** 2) / max(1e-30, np.mean(b.astype(np.float64) ** 2)))


def gpu(sc, W, H, mode, n_timed, flags=0, **ukw):
    ctx = capi.Context(W, H); ctx.set_scene(sc); ctx.set_seed(1); ctx.build_accel(); ctx.set_flags(flags)
    ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, iteration=1, **ukw))
    nb = max(2, ctx.iteration_batch())
    ctx.render(mode, 1, nb, 1); ctx.sync(); ctx.clear_accum(); ctx.reset_stats()
    t0 = time.perf_counter(); ctx.render(mode, 1, n_timed, 1); ctx.sync(); dt = time.perf_counter() - t0
    st = ctx.stats()
    ctx.set_profiling(2); ctx.reset_stats(); ctx.render(mode, 1, 2, 1); ctx.sync(); p = ctx.profile()
    npr = (p.closest_nodes + p.any_nodes) / max(1, p.closest_rays + p.any_rays)
    tpr = (p.closest_tris + p.any_tris) / max(1, p.closest_rays + p.any_rays)
    return ctx, st.rays, dt, st.bvh_build_ms, npr, tpr


def oracle(sc, W, H, mode, iters, flags=0, **ukw):
    o = pyoracle.Oracle(sc, seed=1); o.build_bvh()
    acc = np.zeros((H, W, 4), np.float32); rays = 0
    t0 = time.perf_counter()
    for it in iters:
        _, _, r = o.render(uniforms.make_uniforms(sc, W, H, iteration=it, **ukw), W, H, mode, accum=acc, flags=flags, threads=THREADS)
        rays += r
    return acc, rays, time.perf_counter() - t0


def row(cfg, backend, units, rays, secs, extra=""):
    rows.append(f"| {cfg} | {backend} | {units} | {rays} | {secs:.3f} | {rays / secs / 1e6:.1f} | {extra} |")
    print(rows[-1], flush=True)


# configs[0]: Cornell 512x512 AO 16 rays
sc = scenes.cornell_box()
kw = dict(ao_rays=16, ao_length=1.0)
acc_o, r_o, t_o = oracle(sc, 512, 512, capi.MODE_AO, [1], **kw)
row("Cornell 512² AO 16, 1 iteration", "oracle", f"{THREADS} threads", r_o, t_o)
ctx, r_g, t_g, bms, npr, tpr = gpu(sc, 512, 512, capi.MODE_AO, 256, **kw)
row("Cornell 512² AO 16, 256 iterations", "HIP", "1 GPU", r_g, t_g, f"{npr:.1f} nodes + {tpr:.1f} triangles per ray")
ctx.clear_accum(); ctx.render(capi.MODE_AO, 1, 1, 1); a = ctx.read_accum()
print("   AO image vs oracle: differing floats", int((a.view(np.uint32) != acc_o.view(np.uint32)).sum()), flush=True)
ctx.close()

# configs[1]: conference 1080p, 5 bounces
sc = scenes.conference_room(seed=2, detail=1.4)
kw = dict(bounces=5)
acc_o, r_o, t_o = oracle(sc, 1920, 1080, capi.MODE_PATH, [1, 2], **kw)
row("Conference 1080p B=5, 2 iterations", "oracle", f"{THREADS} threads", r_o, t_o)
ctx, r_g, t_g, bms, npr, tpr = gpu(sc, 1920, 1080, capi.MODE_PATH, 1024, **kw)
row("Conference 1080p B=5, 1024 spp", "HIP", "1 GPU", r_g, t_g, f"{npr:.1f} nodes + {tpr:.1f} triangles per ray; BVH build {bms:.1f} ms")
ctx.clear_accum(); ctx.render(capi.MODE_PATH, 1, 2, 1); a = ctx.read_accum()
print("   2-iteration image vs oracle: differing floats", int((a.view(np.uint32) != acc_o.view(np.uint32)).sum()), flush=True)
ctx.close()

# configs[3]: sponza-scale 1080p, 8 bounces + roulette
sc = scenes.sponza_scale()
kw = dict(bounces=8)
ctx, r_g, t_g, bms, npr, tpr = gpu(sc, 1920, 1080, capi.MODE_PATH, 256, flags=capi.FLAG_RUSSIAN_ROULETTE, **kw)
row("Sponza-scale 1080p B=8 + RR, 256 spp", "HIP", "1 GPU", r_g, t_g, f"{npr:.1f} nodes + {tpr:.1f} triangles per ray; BVH build {bms:.1f} ms")
ctx.close()

if "--big" in sys.argv:
    sc = scenes.san_miguel_scale()
    ctx, r_g, t_g, bms, npr, tpr = gpu(sc, 3840, 2160, capi.MODE_PATH, 32, bounces=5)
    row("San-Miguel-scale 10 M tris 4K B=5, 32 spp", "HIP", "1 GPU", r_g, t_g, f"{npr:.1f} nodes + {tpr:.1f} triangles per ray; BVH build {bms:.1f} ms")
    ctx.close()
print("\n".join(rows))
