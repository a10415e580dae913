"""Randomised GPU-vs-oracle parity: random soups / rooms, sizes, bounces, flags, batches, pipeline depths.
usage: python tools/fuzz_parity.py [seconds] [seed]   -- prints one line per case, exits 1 at the first difference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from pathtracer_amd import capi, scenes, uniforms

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, case = time.time(), 0
while time.time() - t0 < budget:
    case += 1
    kind = rng.integers(0, 4)
    if kind == 0:
        sc = scenes.random_soup(int(rng.integers(1, 6000)), seed=int(rng.integers(1 << 30)), extent=float(rng.uniform(0.5, 8)),
                                size=float(rng.uniform(0.05, 2.0)))
    elif kind == 1:
        sc = scenes.cornell_box()
    elif kind == 2:
        sc = scenes.conference_room(seed=int(rng.integers(1 << 30)), detail=float(rng.uniform(0.15, 0.5)),
                                    textured=bool(rng.integers(0, 2)))
    else:
        sc = scenes.sponza_scale(seed=int(rng.integers(1 << 30)), detail=float(rng.uniform(0.05, 0.2)))
    W, H = int(rng.integers(1, 200)), int(rng.integers(1, 150))
    mode = int(rng.random() < 0.2)
    flags = int(rng.choice([0, 0, capi.FLAG_LITERAL, capi.FLAG_RUSSIAN_ROULETTE, capi.FLAG_LITERAL | capi.FLAG_RUSSIAN_ROULETTE]))
    kw = dict(bounces=int(rng.integers(0, 15)), aa_rays=int(rng.choice([1, 1, 1, 2, 5])), light_area=float(rng.uniform(0.01, 1.0)),
              filter_radius=float(rng.uniform(0, 4)), ao_rays=int(rng.integers(1, 40)), ao_length=float(rng.uniform(0.001, 5)),
              light_e=float(rng.uniform(1, 300)))
    n, first, stride = int(rng.integers(1, 12)), int(rng.integers(1, 50)), int(rng.integers(1, 9))
    seed = int(rng.integers(1 << 31))
    desc = f"case {case}: {sc['name']} {len(sc['indices']) // 3} tris {W}x{H} mode {mode} flags {flags} n {n} first {first} stride {stride} {kw}"
    o = po.Oracle(sc, seed=seed)
    o.build_bvh()
    ctx = capi.Context(W, H)
    ctx.set_scene(sc); ctx.set_seed(seed); ctx.build_accel(); ctx.set_flags(flags)
    ctx.set_pipeline_depth(int(rng.integers(1, 5))); ctx.set_iteration_batch(int(rng.integers(1, 9)))
    ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, **kw))
    ctx.reset_stats()
    ctx.render(mode, first, n, stride)
    acc, ids, rays = np.zeros((H, W, 4), np.float32), None, 0
    for k in range(n):
        acc, ids, r = o.render(uniforms.make_uniforms(sc, W, H, iteration=first + k * stride, **kw), W, H, mode, accum=acc,
                               flags=flags, want_ids=True)
        rays += r
    g = ctx.read_accum()
    nd = int((g.view(np.uint32) != acc.view(np.uint32)).any(axis=-1).sum())
    ok = nd == 0 and (ctx.read_hit_ids() == ids).all() and ctx.stats().rays == rays
    ctx.close()
    print(("ok   " if ok else "DIFF ") + desc + ("" if ok else f" -> {nd} pixels, rays {ctx.stats().rays if False else ''}"), flush=True)
    if not ok:
        sys.exit(1)
print(f"{case} cases, no difference")
