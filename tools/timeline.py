"""Where does a traversal launch's fixed cost go? Per-wave wall-clock stamps of one launch (measurement build
-DPRT_TIMELINE: make variant NAME=timeline DEFS=-DPRT_TIMELINE; PRT_LIB=.../libprt_hip_timeline.so): when each wave of the
persistent grid started, when it found the queue empty, when it left. One-bounce diffuse rays of the bench scene through
prt_trace_device, closest-hit and any-hit, at two launch sizes.  usage: python tools/timeline.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pathtracer_amd import capi, scenes

W, H = 1920, 1080
sc = scenes.conference_room(seed=2, detail=1.4)
ctx = capi.Context(W, H); ctx.set_scene(sc); ctx.build_accel()
cam = sc["camera"]
eye = np.asarray(cam["pos"] if "pos" in cam else cam["eye"], np.float32)
fwd = np.asarray(cam["dir"], np.float32) if "dir" in cam else np.asarray(cam["target"], np.float32) - eye
fwd /= np.linalg.norm(fwd)
right = np.cross(fwd, [0, 1, 0]); right /= np.linalg.norm(right)
up = np.cross(right, fwd)
ys, xs = np.mgrid[0:H, 0:W]
t = np.tan(np.radians(45.0) / 2)
d = (fwd[None, None] + (((xs + 0.5) / W * 2 - 1) * t * W / H)[..., None] * right + ((1 - (ys + 0.5) / H * 2) * t)[..., None] * up)
rays = np.zeros(W * H, capi.RAY_DTYPE)
rays["o"] = eye; rays["d"] = d.reshape(-1, 3).astype(np.float32) * 1000.0; rays["tmin"] = 1e-6; rays["tmax"] = 1.0
hits = ctx.trace(rays)
P = rays["o"] + rays["d"] * hits["t"][:, None]
rng = np.random.default_rng(2)
v = rng.normal(size=(W * H, 3)).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
sec = np.zeros(W * H, capi.RAY_DTYPE)
sec["o"] = P - rays["d"] * 1e-6; sec["d"] = v * 1000.0; sec["tmin"] = 1e-4; sec["tmax"] = 1.0
sec = sec[hits["prim"] != 0xffffffff]
sh = sec.copy(); sh["d"] = sec["d"] * 0.003

read = ctx.L.prt_debug_read_timeline
read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]


def run(name, r, rep, any_hit):
    big = np.concatenate([r] * rep)
    dr = torch.from_numpy(big.view(np.float32).reshape(-1, 8)).cuda()
    dh = torch.empty((len(big), 4), dtype=torch.float32, device="cuda")
    ctx.trace_device(dr.data_ptr(), len(big), dh.data_ptr(), any_hit)
    ms = ctx.trace_device(dr.data_ptr(), len(big), dh.data_ptr(), any_hit)
    tl = np.zeros((8192, 4), np.uint64)
    assert read(ctx.h_, tl.ctypes.data, tl.size) == 0
    tl = tl[tl[:, 2] != 0].astype(np.int64)
    t0 = tl[:, 0].min()
    start, empty, end, n = (tl[:, 0] - t0) / 100.0, (tl[:, 1] - t0) / 100.0, (tl[:, 2] - t0) / 100.0, tl[:, 3]  # microseconds
    q = lambda a: " ".join(f"{x:8.1f}" for x in np.percentile(a, [0, 10, 50, 90, 99, 100]))
    print(f"{name}: {len(big) / 1e6:.1f} M rays, {ms:.3f} ms by events, {len(tl)} waves, rays per wave {n.min()} .. {n.max()}")
    print(f"   percentiles            min      10%      50%      90%      99%      max   (microseconds from the first wave's start)")
    print(f"   wave starts       {q(start)}")
    print(f"   finds queue empty {q(empty)}")
    print(f"   wave leaves       {q(end)}")
    print(f"   empty -> leaves   {q(end - empty)}")
    o = np.argsort(n)
    print("   waves by rays traced, in tenths: rays | finds queue empty | leaves (means)")
    for part in np.array_split(o, 10):
        print(f"      {n[part].mean():7.0f} | {empty[part].mean():8.1f} | {end[part].mean():8.1f}")
    full = np.percentile(empty, 50)
    print(f"   rate until the median wave finds the queue empty: {len(big) / full / 1e3:.2f} Grays/s; launch ends {end.max() - full:.0f} us later; "
          f"waves still running 50 / 100 / 150 / 200 us after that: " + " / ".join(str(int((end > full + x).sum())) for x in (50, 100, 150, 200)), flush=True)


for rep in (2, 8):
    run("closest-hit", sec, rep, False)
    run("any-hit", sh, rep, True)
