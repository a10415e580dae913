"""Nodes / triangles per ray and rate of the bench workload for the library named by PRT_LIB (builder experiments).
usage: PRT_LIB=... python tools/bvh_quality.py [conference|sponza]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import capi, scenes, uniforms
which = sys.argv[1] if len(sys.argv) > 1 else "conference"
sc = scenes.conference_room(seed=2, detail=1.4) if which == "conference" else scenes.sponza_scale(seed=4)
W, H = 1920, 1080
ctx = capi.Context(W, H); ctx.set_scene(sc); ctx.set_seed(2); ctx.build_accel()
st = ctx.stats()
ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=5, iteration=1, light_area=0.02 if which == "conference" else 0.5))
ctx.render(0, 1, 64); ctx.sync(); ctx.reset_stats()
t0 = time.time(); ctx.render(0, 65, 64); ctx.sync(); dt = time.time() - t0
rays = ctx.stats().rays
ctx.set_profiling(2); ctx.reset_stats(); ctx.render(0, 1, 4); p = ctx.profile()
print(f"{os.path.basename(os.environ.get('PRT_LIB', 'libprt_hip.so')):22s} {which}: build {st.bvh_build_ms:6.1f} ms, {st.bvh_inner_nodes} BVH4 nodes depth {st.bvh_depth}, "
      f"closest {p.closest_nodes / p.closest_rays:5.2f} nodes + {p.closest_tris / p.closest_rays:4.2f} tris, "
      f"any {p.any_nodes / p.any_rays:5.2f} + {p.any_tris / p.any_rays:4.2f}, {rays / dt / 1e6:7.1f} Mrays/s", flush=True)
