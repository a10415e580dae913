import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd import scenes, uniforms, capi
def P(*a): print(*a, flush=True)
detail = float(sys.argv[1]); W = int(sys.argv[2]); H = int(sys.argv[3]); prof = int(sys.argv[4]); B = int(sys.argv[5])
sc = scenes.conference_room(detail=detail)
P("tris", len(sc['indices'])//3)
ctx = capi.Context(W, H); ctx.set_scene(sc); ctx.set_seed(2); ctx.build_accel()
s = ctx.stats(); P("bvh", s.bvh_nodes, s.bvh_depth, s.bvh_build_ms)
ctx.set_profiling(prof)
for it in range(1, 4):
    u = uniforms.make_uniforms(sc, W, H, bounces=B, iteration=it)
    ctx.set_uniforms(u); ctx.reset_stats(); t = time.time(); ctx.render_iteration(0); ctx.sync(); dt = time.time() - t
    P("iter", it, "rays", ctx.stats().rays, "ms", dt * 1e3, "Mrays/s", ctx.stats().rays / dt / 1e6)
if prof:
    p = ctx.profile()
    for k, n in capi.KERNEL_NAMES.items():
        if p.launches[k]: P(n, p.ms[k], p.launches[k])
    P("closest nodes/ray", p.closest_nodes / max(1, p.closest_rays), "tris/ray", p.closest_tris / max(1, p.closest_rays))
    P("any nodes/ray", p.any_nodes / max(1, p.any_rays), "tris/ray", p.any_tris / max(1, p.any_rays))
