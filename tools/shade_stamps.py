"""Where k_shade's cycles go (measurement build libprt_hip_stamps.so, -DPRT_SHADE_STAMPS). usage: PRT_LIB=... python tools/shade_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd import capi, scenes, uniforms
W, H = 1920, 1080
sc = scenes.conference_room(seed=2, detail=1.4)
ctx = capi.Context(W, H); ctx.set_scene(sc); ctx.set_seed(2); ctx.build_accel()
ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=5, iteration=1))
ctx.set_pipeline_depth(1)
ctx.render(0, 1, 16, 1); ctx.sync()
ctx.reset_stats(); ctx.set_profiling(1)
ctx.render(0, 17, 16, 1)
p = ctx.profile()
st = (ctypes.c_ulonglong * 48)()
ctx.L.prt_debug_read_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
ctx._chk(ctx.L.prt_debug_read_stats(ctx.h_, st))
t = np.array(st[36:42], np.float64); trips = st[42]
names = ["queue slot + path state", "shading triangle", "material/textures/light/BSDF", "stores + termination", "compaction 1", "compaction 2"]
print(f"k_shade: {p.ms[capi.K_SHADE] / p.launches[capi.K_SHADE]:.3f} ms per launch (stamped build, serialised waits), block trips {trips}")
for n, v in zip(names, t):
    print(f"   {n:32s} {v / t.sum():6.3f}   {v / (trips * 4):9.0f} cycles per wave-trip")
print(f"   total {t.sum() / (trips * 4):.0f} cycles per wave-trip")
