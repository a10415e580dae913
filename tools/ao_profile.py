"""Per-kernel times of the Cornell AO configuration (BASELINE.json configs[0]): python tools/ao_profile.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import capi, scenes, uniforms
sc = scenes.cornell_box()
W = H = 512
ctx = capi.Context(W, H); ctx.set_scene(sc); ctx.set_seed(1); ctx.build_accel()
ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, ao_rays=16, ao_length=1.0, iteration=1))
ctx.render(capi.MODE_AO, 1, 64); ctx.sync(); ctx.reset_stats()
t0 = time.time(); ctx.render(capi.MODE_AO, 65, 256); ctx.sync(); dt = time.time() - t0
print("rays", ctx.stats().rays, "s", dt, "Mrays/s", ctx.stats().rays / dt / 1e6, "batch", ctx.iteration_batch())
ctx.set_profiling(1); ctx.reset_stats(); ctx.render(capi.MODE_AO, 1, 256); ctx.sync(); p = ctx.profile()
for n in [n for n in dir(capi) if n.startswith("K_")]:
    k = getattr(capi, n)
    if isinstance(k, int) and k < len(p.launches) and p.launches[k]:
        print(n, p.launches[k], "launches", round(p.ms[k], 3), "ms total")
ctx.close()
