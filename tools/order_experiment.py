"""Does the order of rays in the queue matter? Primary and one-bounce diffuse rays of the bench scene, traced through
prt_trace_device in row-major pixel order (what a pass does), in 8x8 pixel tiles, and shuffled. Measurement tool."""
import os, sys
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
dx = ((xs + 0.5) / W * 2 - 1) * t * W / H
dy = (1 - (ys + 0.5) / H * 2) * t
d = (fwd[None, None] + dx[..., None] * right + dy[..., None] * up).astype(np.float32) * 1000.0
rays = np.zeros(W * H, capi.RAY_DTYPE)
rays["o"] = eye; rays["d"] = d.reshape(-1, 3); rays["tmin"] = 1e-6; rays["tmax"] = 1.0

def orders():
    lin = np.arange(W * H)
    tile = ((ys // 8) * (W // 8) + xs // 8) * 64 + (ys % 8) * 8 + xs % 8
    o_tile = np.argsort(tile.reshape(-1), kind="stable")
    rng = np.random.default_rng(1)
    return {"row-major": lin, "8x8 tiles": o_tile, "shuffled": rng.permutation(W * H)}

def run(name, r, any_hit):
    rep = 8
    big = np.concatenate([r] * rep)  # 16.6 M rays: a launch as long as the bench's
    dr = torch.from_numpy(big.view(np.float32).reshape(-1, 8)).cuda()
    dh = torch.empty((len(big), 4), dtype=torch.float32, device="cuda")
    ctx.trace_device(dr.data_ptr(), len(big), dh.data_ptr(), any_hit)
    ms = min(ctx.trace_device(dr.data_ptr(), len(big), dh.data_ptr(), any_hit) for _ in range(3))
    print(f"   {name:12s} {ms:7.3f} ms  {len(big) / ms / 1e6:7.2f} Grays/s", flush=True)
    return dh[:len(r)].cpu().numpy()

od = orders()
print("primary rays (closest hit)")
hits = None
for n, o in od.items():
    h = run(n, rays[o], False)
    if n == "row-major": hits = h
# one diffuse bounce from the primary hit points
tt = hits[:, 3:4]
P = rays["o"] + rays["d"] * tt
rng = np.random.default_rng(2)
v = rng.normal(size=(W * H, 3)).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
sec = np.zeros(W * H, capi.RAY_DTYPE)
sec["o"] = P - rays["d"] * 1e-6; sec["d"] = v * 1000.0; sec["tmin"] = 1e-4; sec["tmax"] = 1.0
ok = hits.view(np.uint32)[:, 0] != 0xffffffff
print("diffuse bounce rays (closest hit)")
for n, o in od.items():
    run(n, sec[o][ok[o]], False)
print("diffuse bounce rays (any hit, tmax 0.003 = 3 m)")
sh = sec.copy(); sh["d"] = v * 3.0
for n, o in od.items():
    run(n, sh[o][ok[o]], True)
# sorted by the Morton code of the origin (what a sort before traversal could give at best)
q = ((P - P[ok].min(0)) / (np.ptp(P[ok], axis=0) + 1e-6) * 1023).clip(0, 1023).astype(np.uint64)
def part(x):
    x = (x | (x << 32)) & 0x1f00000000ffff; x = (x | (x << 16)) & 0x1f0000ff0000ff
    x = (x | (x << 8)) & 0x100f00f00f00f00f; x = (x | (x << 4)) & 0x10c30c30c30c30c3
    return (x | (x << 2)) & 0x1249249249249249
m = part(q[:, 0]) | (part(q[:, 1]) << 1) | (part(q[:, 2]) << 2)
om = np.argsort(m, kind="stable")
print("diffuse bounce rays, sorted by origin Morton code")
run("closest", sec[om][ok[om]], False)
run("any", sh[om][ok[om]], True)
