"""Randomised GPU-vs-oracle parity: random soups / rooms, sizes, bounces, flags, batches, pipeline depths, pass-memory
limits, and a second render after the scene moved (instance placements / vertex jitter + refit).
usage: python tools/fuzz_parity.py [seconds] [seed] [big]   -- prints one line per case, exits 1 at the first difference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from pathtracer_amd import capi, scenes, uniforms

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
big = len(sys.argv) > 3 and sys.argv[3] == "big"  # fewer, larger cases: up to 640x400 pixels, scenes up to ~150k triangles
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, case = time.time(), 0
while time.time() - t0 < budget:
    case += 1
    kind = rng.integers(0, 7)
    if kind >= 5:  # two-level scene: instances with random transforms (prt_set_instances)
        sc = scenes.instanced_scene(seed=int(rng.integers(1 << 30)), n_instances=int(rng.integers(1, 400 if big else 60)),
                                    detail=float(rng.uniform(0.3, 2.0)), wild=bool(rng.integers(0, 2)))
    elif kind == 0:
        sc = scenes.random_soup(int(rng.integers(1, 6000)), seed=int(rng.integers(1 << 30)), extent=float(rng.uniform(0.5, 8)),
                                size=float(rng.uniform(0.05, 2.0)))
    elif kind == 1:
        sc = scenes.cornell_box()
    elif kind == 2:
        sc = scenes.conference_room(seed=int(rng.integers(1 << 30)), detail=float(rng.uniform(0.5, 1.0) if big else rng.uniform(0.15, 0.5)),
                                    textured=bool(rng.integers(0, 2)))
    elif kind == 4:  # soup with duplicates, zero-area triangles and a far copy
        base = scenes.random_soup(int(rng.integers(3, 800)), seed=int(rng.integers(1 << 30)), size=float(rng.uniform(0.01, 1.0)))
        P = base["vertices"]["p"][base["indices"]].reshape(-1, 3, 3).astype(np.float32)
        m = max(1, len(P) // 4)
        mb = scenes.MeshBuilder()
        mb.add(np.concatenate([P, P[:m], np.stack([P[:m, 0], P[:m, 0], P[:m, 1]], axis=1),
                               np.stack([P[:m, 0], (P[:m, 0] + P[:m, 1]) * 0.5, P[:m, 1]], axis=1),
                               P[:m] + np.float32(rng.uniform(100, 5000))]), 0)
        sc = mb.finish([scenes.make_material()])
        sc["name"], sc["camera"], sc["light"] = "degenerate", base["camera"], base["light"]
    else:
        sc = scenes.sponza_scale(seed=int(rng.integers(1 << 30)), detail=float(rng.uniform(0.3, 0.7) if big else rng.uniform(0.05, 0.2)))
    W, H = (int(rng.integers(200, 640)), int(rng.integers(150, 400))) if big else (int(rng.integers(1, 200)), int(rng.integers(1, 150)))
    mode = int(rng.random() < 0.2)
    flags = int(rng.choice([0, 0, capi.FLAG_LITERAL, capi.FLAG_RUSSIAN_ROULETTE, capi.FLAG_LITERAL | capi.FLAG_RUSSIAN_ROULETTE]))
    kw = dict(bounces=int(rng.integers(0, 15)), aa_rays=int(rng.choice([1, 1, 1, 2, 5])), light_area=float(rng.uniform(0.01, 1.0)),
              filter_radius=float(rng.uniform(0, 4)), ao_rays=int(rng.integers(1, 40)), ao_length=float(rng.uniform(0.001, 5)),
              light_e=float(rng.uniform(1, 300)))
    n, first, stride = int(rng.integers(1, 12)), int(rng.integers(1, 50)), int(rng.integers(1, 9))
    seed = int(rng.integers(1 << 31))
    desc = f"case {case}: {sc['name']} {len(sc['indices']) // 3} tris {W}x{H} mode {mode} flags {flags} n {n} first {first} stride {stride} {kw}"
    depth, batch = int(rng.integers(1, 5)), int(rng.integers(1, 9))
    desc += f" depth {depth} batch {batch} seed {seed}"
    only = int(os.environ.get("FUZZ_ONLY", "0"))  # replay: run only this case of the sequence
    if only and case != only:
        if case > only:
            break
        continue
    if os.environ.get("FUZZ_DEPTH"):
        depth, batch = int(os.environ["FUZZ_DEPTH"]), int(os.environ.get("FUZZ_BATCH", batch))
    if os.environ.get("FUZZ_DUMP_SCENE"):  # for CPU-side debugging of a case found on the GPU box
        import pickle
        pickle.dump(dict(scene=sc, W=W, H=H, mode=mode, flags=flags, kw=kw, n=n, first=first, stride=stride, seed=seed),
                    open(os.environ["FUZZ_DUMP_SCENE"], "wb"))
        print("dumped", desc)
        sys.exit(0)
    table = None
    if rng.random() < 0.2:  # the reference's explicit 32-layer scramble image instead of the hash
        table = rng.integers(0, 2 ** 32, (32, H * W), dtype=np.uint64).astype(np.uint32)

    def make_oracle(scene):
        o = po.Oracle(scene, seed=seed)
        if "instances" in scene:
            o.set_instances(scene["blas"], scene["instances"])
        o.build_bvh()
        if table is not None:
            o.set_scramble_table(table, W, H)
        return o

    def compare(ctx, o):
        ctx.clear_accum(); ctx.reset_stats()
        ctx.render(mode, first, n, stride)
        acc, ids, rays = np.zeros((H, W, 4), np.float32), None, 0
        for k in range(n):
            acc, ids, r = o.render(uniforms.make_uniforms(sc, W, H, iteration=first + k * stride, **kw), W, H, mode, accum=acc,
                                   flags=flags, want_ids=True)
            rays += r
        g = ctx.read_accum()
        nd = int((g.view(np.uint32) != acc.view(np.uint32)).any(axis=-1).sum())
        gids, st = ctx.read_hit_ids(), ctx.stats()
        return (nd == 0 and (gids == ids).all() and st.rays == rays and st.trace_overruns == 0), nd, g, acc, gids, ids, st.rays, rays

    o = make_oracle(sc)
    ctx = capi.Context(W, H)
    ctx.set_scene(sc)
    if "instances" in sc:
        ctx.set_instances(sc["blas"], sc["instances"])
    ctx.set_seed(seed); ctx.build_accel(); ctx.set_flags(flags)
    ctx.set_pipeline_depth(depth); ctx.set_iteration_batch(batch)
    if rng.random() < 0.25:  # memory policy: a limit that lets only part of the batch through
        ctx.set_pass_memory_limit(int(W * H * (218 + 16 * kw["bounces"]) * rng.uniform(0.5, batch)))
        desc += " limited"
    if table is not None:
        ctx.upload_scrambles(table)
    ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, **kw))
    ok, nd, g, acc, gids, ids, grays, rays = compare(ctx, o)
    if "instances" in sc:  # random rays against the oracle's brute force over all instances
        nr = 20000
        arr = np.zeros(nr, capi.RAY_DTYPE)
        arr["o"] = rng.uniform(-6, 6, (nr, 3)); arr["d"] = rng.normal(0, 1, (nr, 3)) * rng.uniform(0.01, 10, (nr, 1))
        arr["tmin"], arr["tmax"] = 1e-6, rng.choice([0.5, 2.0, 1e3], nr)
        ok = ok and ctx.trace(arr).tobytes() == o.trace_closest(arr, brute=True).tobytes()
        ok = ok and ((ctx.trace(arr, any_hit=True)["prim"] != 0xFFFFFFFF) == o.trace_any(arr, brute=True).astype(bool)).all()
    if ok and rng.random() < 0.4:
        # the scene moves: instances get new placements (top level rebuilt) or the vertices are jittered (refit);
        # the same context renders again and must equal the oracle on the moved scene
        if "instances" in sc:
            inst = sc["instances"].copy()
            for i in range(len(inst)):
                # column-major storage: the flat arrays are the transposes, (D * M)^T = M^T * D^T
                d = scenes.affine(translate=tuple(rng.normal(0, 0.5, 3)), axis=tuple(rng.normal(0, 1, 3) + 1e-3),
                                  angle=float(rng.uniform(-1, 1)), scale=float(rng.uniform(0.7, 1.4))).reshape(4, 4)
                inst["transform"][i] = (inst["transform"][i].reshape(4, 4) @ d).astype(np.float32).reshape(-1)
            sc = dict(sc, instances=inst)
            ctx.update_instance_transforms(inst["transform"])
            desc += " +moved"
        else:
            v = sc["vertices"].copy()
            v["p"] += rng.normal(0, 0.01 * float(np.abs(v["p"]).max()) * rng.uniform(0, 1), v["p"].shape).astype(np.float32)
            sc = dict(sc, vertices=v)
            ctx.update_vertices(v); ctx.refit_accel()
            desc += " +refit"
        o = make_oracle(sc)
        ok, nd, g, acc, gids, ids, grays, rays = compare(ctx, o)
    if not ok and os.environ.get("FUZZ_DEBUG"):
        # one iteration at a time: which iteration, which pixels, how many rays
        for k in range(n):
            u = uniforms.make_uniforms(sc, W, H, iteration=first + k * stride, **kw)
            fpos = os.path.getsize(os.environ["FUZZ_RAYS"]) if os.environ.get("FUZZ_RAYS") else 0
            a1, i1, r1 = o.render(u, W, H, mode, flags=flags, want_ids=True)
            ctx.clear_accum(); ctx.set_uniforms(u); ctx.reset_stats(); ctx.render_iteration(mode)
            g1 = ctx.read_accum()
            d1 = np.argwhere((g1.view(np.uint32) != a1.view(np.uint32)).any(axis=-1))
            print(f"   iteration {first + k * stride}: {len(d1)} pixels differ, ids differ {int((ctx.read_hit_ids() != i1).sum())}, rays {ctx.stats().rays} vs {r1}")
            for y, x in d1[:3]:
                print(f"      ({x},{y}) gpu {g1[y, x]} oracle {a1[y, x]} prim {i1[y, x]}")
            if len(d1) and os.environ.get("ORC_DEBUG_PIXEL"):
                # the oracle printed the pixel's rays on stderr (captured in FUZZ_RAYS by the caller): trace them on the GPU
                import re
                rays = []
                fh = open(os.environ["FUZZ_RAYS"])
                fh.seek(fpos)
                for line in fh:
                    m = re.match(r"ORC (\w+) aa (\d+) bounce (\d+) o (\S+) (\S+) (\S+) tmin (\S+) d (\S+) (\S+) (\S+) tmax (\S+) -> prim (\d+) t (\S+)", line)
                    if m:
                        rays.append((m.group(1), [float.fromhex(v) for v in m.groups()[3:11]], int(m.group(12)), m.group(13)))
                rays = rays[-64:]
                arr = np.zeros(len(rays), capi.RAY_DTYPE)
                for i, (_, v, _, _) in enumerate(rays):
                    arr["o"][i], arr["tmin"][i], arr["d"][i], arr["tmax"][i] = v[0:3], v[3], v[4:7], v[7]
                hc, ha = ctx.trace(arr), ctx.trace(arr, any_hit=True)
                oc = o.trace_closest(arr, brute=True)
                for i, (what, v, prim, t) in enumerate(rays):
                    print(f"      {what:8s} oracle prim {prim} t {t} | gpu closest {hc['prim'][i]} t {float(hc['t'][i]).hex()} any {ha['prim'][i]} | brute {oc['prim'][i]} {float(oc['t'][i]).hex()} | o {v[0:3]} d {v[4:7]}")
            sys.stderr.flush()
    ctx.close()
    print(("ok   " if ok else "DIFF ") + desc + ("" if ok else f" -> {nd} pixels differ, {int((gids != ids).sum())} hit ids differ, "
                                                               f"rays {grays} vs {rays}"), flush=True)
    if not ok and nd:
        bad = np.argwhere((g.view(np.uint32) != acc.view(np.uint32)).any(axis=-1))[:5]
        for y, x in bad:
            print(f"   pixel ({x},{y}): gpu {g[y, x]} oracle {acc[y, x]}", flush=True)
    if not ok:
        sys.exit(1)
print(f"{case} cases, no difference")
