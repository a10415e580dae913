"""What would splitting large triangles buy? An architectural scene with LARGE triangles (walls of two triangles, long
diagonal beams) and fine clutter, traced (a) as it is and (b) with every triangle cut along its longest edge until no
edge is longer than L -- the geometry is the same, the boxes are what reference splitting would give at best.
Measurement tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd import capi, scenes, uniforms


def coarse_architecture(seed=9):
    rng = np.random.default_rng(seed)
    mats = [scenes.make_material((0.7, 0.7, 0.7), roughness=0.9), scenes.make_material((0.6, 0.3, 0.2), (0.1, 0.1, 0.1), roughness=0.4)]
    mb = scenes.MeshBuilder()
    L, Wd, Ht = 30.0, 12.0, 14.0
    x0, x1, z0, z1 = -L / 2, L / 2, -Wd / 2, Wd / 2
    mb.quad_grid((x0, 0, z0), (0, 0, Wd), (L, 0, 0), 1, 1, 0)
    mb.quad_grid((x0, Ht, z0), (L, 0, 0), (0, 0, Wd), 1, 1, 0)
    mb.quad_grid((x0, 0, z0), (L, 0, 0), (0, Ht, 0), 1, 1, 0)
    mb.quad_grid((x0, 0, z1), (0, Ht, 0), (L, 0, 0), 1, 1, 0)
    mb.quad_grid((x0, 0, z0), (0, Ht, 0), (0, 0, Wd), 1, 1, 0)
    mb.quad_grid((x1, 0, z0), (0, 0, Wd), (0, Ht, 0), 1, 1, 0)
    for zc in (z0 + 1.5, z1 - 1.5):
        mb.box((x0, 5.0, zc - 1.5), (x1, 5.3, zc + 1.5), 0, n=1)
    for zc in (z0 + 3.0, z1 - 3.0):
        for xc in np.linspace(x0 + 2.0, x1 - 2.0, 11):
            mb.cylinder((xc, 0.0, zc), 0.32, 5.0, 24, 0, rings=1)
            mb.cylinder((xc, 5.3, zc), 0.32, 4.2, 24, 0, rings=1)
    for _ in range(40):  # long beams at odd angles: the boxes of their triangles are mostly empty
        c = np.array([rng.uniform(x0 + 5, x1 - 5), rng.uniform(6, 13), rng.uniform(z0 + 2, z1 - 2)])
        mb.box(c - (4.0, 0.05, 0.05), c + (4.0, 0.05, 0.05), 1, n=1, rot_y=rng.uniform(0, 3.14))
    for k, xc in enumerate(np.linspace(x0 + 4.0, x1 - 4.0, 6)):
        ph = rng.uniform(0, 6.28)
        wave = lambda S, T, ph=ph: np.stack([0.25 * np.sin(9.0 * T + ph) * S, np.zeros_like(S), 0.15 * np.sin(7.0 * S + ph)], -1)
        mb.quad_grid((xc, 9.5, -2.2), (0, -6.0, 0), (0, 0, 4.4), 70, 60, 1, displace=wave)
    for _ in range(200):
        c = (rng.uniform(x0 + 1, x1 - 1), rng.uniform(0.2, 4.0), rng.uniform(z0 + 1, z1 - 1))
        mb.sphere(c, rng.uniform(0.08, 0.3), 32, 16, 1)
    sc = mb.finish(mats)
    sc["name"] = "coarse_architecture"
    sc["camera"] = scenes._lookat_pose((x0 + 2.5, 2.2, 0.6), (x1, 5.0, -0.5))
    sc["light"] = {"pos": (0.0, Ht - 0.5, 0.0), "dir": (0.0, -1.0, 0.0)}
    return sc


def subdivide(sc, maxlen):
    v = sc["vertices"]
    P = v["p"].reshape(-1, 3, 3).astype(np.float64)
    M = v["materialID"].reshape(-1, 3)[:, 0]
    out_p, out_m = [], []
    while len(P):
        e = np.linalg.norm(P[:, [1, 2, 0]] - P, axis=2)  # edge k: vertex k -> k+1
        big = e.max(1) > maxlen
        out_p.append(P[~big]); out_m.append(M[~big])
        P, M, e = P[big], M[big], e[big]
        if not len(P): break
        k = e.argmax(1); i = np.arange(len(P))
        a, b, c = P[i, k], P[i, (k + 1) % 3], P[i, (k + 2) % 3]
        m = 0.5 * (a + b)
        P = np.concatenate([np.stack([a, m, c], 1), np.stack([m, b, c], 1)]); M = np.concatenate([M, M])
    mb = scenes.MeshBuilder()
    P, M = np.concatenate(out_p), np.concatenate(out_m)
    for mat in np.unique(M): mb.add(P[M == mat], int(mat))
    s2 = mb.finish(sc["materials"])
    for k in ("name", "camera", "light"): s2[k] = sc[k]
    return s2


def measure(sc, label, W=1920, H=1080):
    ctx = capi.Context(W, H); ctx.set_scene(sc); ctx.set_seed(2); ctx.build_accel()
    ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=5, iteration=1))
    ctx.set_profiling(2); ctx.reset_stats(); ctx.render(0, 1, 2, 1); ctx.sync()
    p = ctx.profile()
    ctx.set_profiling(0); ctx.render(0, 3, 16, 1); ctx.sync(); ctx.reset_stats()
    t0 = time.perf_counter(); ctx.render(0, 19, 64, 1); ctx.sync(); dt = time.perf_counter() - t0
    rays = ctx.stats().rays
    print(f"{label:34s} {len(sc['indices']) // 3:8d} tris  closest {p.closest_nodes / p.closest_rays:6.2f} nodes {p.closest_tris / p.closest_rays:5.2f} tris per ray"
          f"   any {p.any_nodes / p.any_rays:6.2f} / {p.any_tris / p.any_rays:5.2f}   {rays / dt / 1e6:7.0f} Mrays/s", flush=True)
    ctx.close()


if __name__ == "__main__":
    base = coarse_architecture()
    measure(base, "coarse architecture, as modelled")
    for L in (4.0, 2.0, 1.0, 0.5):
        measure(subdivide(base, L), f"  edges cut to <= {L} m")
    sp = scenes.sponza_scale()
    measure(sp, "sponza-scale stand-in")
    measure(subdivide(sp, 0.15), "  edges cut to <= 0.15 m")
