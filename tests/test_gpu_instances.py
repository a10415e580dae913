"""GPU parity tests of two-level scenes (prt_set_instances: instances of bottom-level structures with transforms,
/root/reference/external/NVIDIA_RTX/vkRT_TLAS.cpp:58-72) against the CPU oracle's definition (object-space
intersection, oracle.cpp "instances"; PARITY UNPINNED like the single-level trace). Bar: hits, primary ids, ray counts
and the fp32 image bit-exact."""
import os
import sys

import numpy as np
import pytest

from pathtracer_amd import capi, scenes, uniforms

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import instancing  # noqa: E402

pytestmark = pytest.mark.gpu


def _pair(po, sc, W, H, seed=3, flags=0):
    o = po.Oracle(sc, seed=seed)
    o.set_instances(sc["blas"], sc["instances"])
    o.build_bvh()
    ctx = capi.Context(W, H)
    ctx.set_scene(sc)
    ctx.set_instances(sc["blas"], sc["instances"])
    ctx.set_seed(seed)
    ctx.build_accel()
    ctx.set_flags(flags)
    return o, ctx


def _rays(n, seed, tmax=100.0, nan_every=0):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, capi.RAY_DTYPE)
    rays["o"] = rng.uniform(-4.5, 4.5, (n, 3))
    rays["o"][:, 1] = rng.uniform(0.1, 3.9, n)
    rays["d"] = rng.normal(0, 1, (n, 3)) * rng.uniform(0.2, 3.0, (n, 1))
    # some axis-parallel and some far-away rays
    rays["d"][::17, 0] = 0.0
    rays["d"][::29, 1:] = 0.0
    rays["o"][::23] *= 40.0
    rays["tmin"] = 1e-6
    rays["tmax"] = tmax
    if nan_every:
        rays["d"][::nan_every, 2] = np.nan
    return rays


@pytest.mark.parametrize("wild,n_inst,seed", [(False, 24, 11), (True, 24, 12), (True, 300, 13), (True, 1, 14)])
def test_instanced_trace_equals_oracle_brute_force(po, wild, n_inst, seed):
    sc = scenes.instanced_scene(seed=seed, n_instances=n_inst, wild=wild)
    o, ctx = _pair(po, sc, 16, 16)
    rays = _rays(60000, seed, nan_every=997)
    g = ctx.trace(rays)
    ref = o.trace_closest(rays, brute=True)
    assert g.tobytes() == ref.tobytes(), f"{int((g['prim'] != ref['prim']).sum())} closest hits differ"
    short = _rays(60000, seed + 100, tmax=1.0)
    ga = ctx.trace(short, any_hit=True)["prim"] != 0xFFFFFFFF
    ra = o.trace_any(short, brute=True).astype(bool)
    assert (ga == ra).all() and 0.02 < ra.mean() < 0.98
    assert ctx.stats().trace_overruns == 0
    ctx.close()


@pytest.mark.parametrize("mode,bounces,flags", [(0, 3, 0), (0, 5, capi.FLAG_RUSSIAN_ROULETTE), (1, 0, 0),
                                                (0, 2, capi.FLAG_LITERAL)])
def test_instanced_image_bit_exact(po, mode, bounces, flags):
    sc = scenes.instanced_scene(seed=15, n_instances=20, wild=True)
    W, H = 96, 64
    o, ctx = _pair(po, sc, W, H, flags=flags)
    acc = np.zeros((H, W, 4), np.float32)
    for it in (1, 2, 3):
        u = uniforms.make_uniforms(sc, W, H, iteration=it, bounces=bounces, ao_rays=8)
        acc, ids, rays = o.render(u, W, H, mode, accum=acc, flags=flags, want_ids=True)
        ctx.set_uniforms(u)
        ctx.reset_stats()
        ctx.render_iteration(mode)
        g = ctx.read_accum()
        assert (ctx.read_hit_ids() == ids).all(), f"iteration {it}: primary hit ids differ"
        assert ctx.stats().rays == rays
        nd = int((g.view(np.uint32) != acc.view(np.uint32)).any(axis=-1).sum())
        assert nd == 0, f"iteration {it}: {nd} pixels differ"
    ctx.close()


def test_instanced_batched_render_bit_exact(po):
    """prt_render (several iterations per pass, pipelined) on a two-level scene = the oracle iteration by iteration."""
    sc = scenes.instanced_scene(seed=16, n_instances=30, wild=False)
    W, H = 80, 60
    o, ctx = _pair(po, sc, W, H)
    u = uniforms.make_uniforms(sc, W, H, iteration=1, bounces=4)
    ctx.set_uniforms(u)
    ctx.set_iteration_batch(4)
    ctx.render(capi.MODE_PATH, 1, 7)
    g = ctx.read_accum()
    acc = np.zeros((H, W, 4), np.float32)
    for it in range(1, 8):
        acc, _, _ = o.render(uniforms.make_uniforms(sc, W, H, iteration=it, bounces=4), W, H, 0, accum=acc)
    assert g.tobytes() == acc.tobytes()
    # AO mode batches the same way (k_ao_shade<TWO, BATCH>)
    ctx.clear_accum()
    ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, iteration=1, ao_rays=6, ao_length=0.7))
    ctx.reset_stats()
    ctx.render(capi.MODE_AO, 1, 5)
    acc, rays = np.zeros((H, W, 4), np.float32), 0
    for it in range(1, 6):
        acc, ids, r = o.render(uniforms.make_uniforms(sc, W, H, iteration=it, ao_rays=6, ao_length=0.7), W, H, 1, accum=acc,
                               want_ids=True)
        rays += r
    assert ctx.read_accum().tobytes() == acc.tobytes() and (ctx.read_hit_ids() == ids).all() and ctx.stats().rays == rays
    ctx.close()


def test_identity_instance_equals_single_level_context(po):
    """What the reference wires — one identity instance of model 0 (vkRTX_setup.cpp:138-186) — gives the bits of the
    single-level path."""
    sc = scenes.conference_room(detail=0.3)
    W, H = 64, 48
    u = uniforms.make_uniforms(sc, W, H, iteration=1, bounces=3)
    imgs, ids = [], []
    for two in (False, True):
        ctx = capi.Context(W, H)
        ctx.set_scene(sc)
        if two:
            inst = np.zeros(1, capi.INSTANCE_DTYPE)
            inst["transform"][0] = scenes.affine()
            ctx.set_instances(np.array([(0, len(sc["indices"]) // 3)], capi.BLAS_DTYPE), inst)
        ctx.set_seed(3)
        ctx.build_accel()
        ctx.set_uniforms(u)
        ctx.render(capi.MODE_PATH, 1, 3)
        imgs.append(ctx.read_accum())
        ids.append(ctx.read_hit_ids())
        ctx.close()
    assert imgs[0].tobytes() == imgs[1].tobytes() and (ids[0] == ids[1]).all()


def test_instanced_close_to_flattened(po):
    """Against the host-flattened scene (pathtracer_amd/host/Scene.cpp semantics, world-space triangles): the same
    primitives up to grazing cases and the same image up to the rounding of the object-space transforms."""
    sc = scenes.instanced_scene(seed=17, n_instances=16, wild=True)
    W, H = 96, 64
    u = uniforms.make_uniforms(sc, W, H, iteration=1, bounces=3)
    ctx = capi.Context(W, H)
    ctx.set_scene(sc)
    ctx.set_instances(sc["blas"], sc["instances"])
    ctx.build_accel()
    ctx.set_uniforms(u)
    ctx.render(capi.MODE_PATH, 1, 4)
    a, ia = ctx.read_accum(), ctx.read_hit_ids()
    nodes_two = ctx.stats().bvh_inner_nodes
    ctx.close()
    ctx = capi.Context(W, H)
    ctx.set_scene(instancing.flatten(sc))
    ctx.build_accel()
    ctx.set_uniforms(u)
    ctx.render(capi.MODE_PATH, 1, 4)
    b, ib = ctx.read_accum(), ctx.read_hit_ids()
    nodes_flat = ctx.stats().bvh_inner_nodes
    ctx.close()
    assert (ia == ib).mean() > 0.999
    assert float(((a - b) ** 2).sum() / (b ** 2).sum()) < 1e-6
    assert nodes_two < nodes_flat  # the structures are shared, not copied per instance


def test_blas_trees_are_the_oracle_trees(po):
    sc = scenes.instanced_scene(seed=18, n_instances=5)
    o, ctx = _pair(po, sc, 16, 16)
    for k in range(len(sc["blas"])):
        gn, go = ctx.read_blas_bvh(k)
        on, oo = o.blas_bvh(k)
        assert gn.tobytes() == on.tobytes() and (go == oo).all()
    ctx.close()


def test_instance_transform_update_equals_fresh_build(po):
    sc = scenes.instanced_scene(seed=19, n_instances=40, wild=True)
    sc2 = scenes.instanced_scene(seed=20, n_instances=40, wild=True)
    sc2["instances"]["blas"] = sc["instances"]["blas"]  # same structures, new placements
    o, ctx = _pair(po, sc, 16, 16)
    rays = _rays(30000, 5)
    ctx.update_instance_transforms(sc2["instances"]["transform"])
    g = ctx.trace(rays)
    o2 = po.Oracle(sc, seed=3)
    o2.set_instances(sc["blas"], sc2["instances"])
    assert g.tobytes() == o2.trace_closest(rays, brute=True).tobytes()
    ctx.close()


def test_refused_transform_update_leaves_the_scene_as_it_was(po):
    """prt_update_instance_transforms checks every transform before it commits any: after a refused update (singular,
    not finite) the context still traces the old placements. Several updates in a row (the top level is rebuilt each
    time, the bottom-level structures' nodes and leaf records stay) end where a fresh build of the last one ends,
    also when the top level shrinks and grows between them."""
    sc = scenes.instanced_scene(seed=23, n_instances=30, wild=True)
    o, ctx = _pair(po, sc, 16, 16)
    rays = _rays(20000, 8)
    before = ctx.trace(rays).tobytes()
    assert before == o.trace_closest(rays, brute=True).tobytes()
    for bad_value in (scenes.affine(scale=(1.0, 0.0, 1.0)), np.full(16, np.nan, np.float32)):
        t = sc["instances"]["transform"].copy()
        t[7] = np.asarray(bad_value, np.float32).reshape(-1)
        if np.isnan(t[7]).any():
            t[7][[3, 7, 11]] = 0.0
            t[7][15] = 1.0  # a legal fourth row, so that the finiteness check is what refuses it
        with pytest.raises(capi.PrtError):
            ctx.update_instance_transforms(t)
        assert ctx.trace(rays).tobytes() == before
    # far apart -> all in one spot -> the placements of another scene: top-level trees of different sizes
    spread = sc["instances"]["transform"].copy()
    for i in range(len(spread)):
        spread[i] = np.asarray(scenes.affine(translate=(7.0 * i, 0.0, -3.0 * i)), np.float32).reshape(-1)
    heap = sc["instances"]["transform"].copy()
    heap[:] = heap[0]
    sc2 = scenes.instanced_scene(seed=24, n_instances=30, wild=True)
    for t in (spread, heap, sc2["instances"]["transform"]):
        ctx.update_instance_transforms(t)
        inst = sc["instances"].copy()
        inst["transform"] = t
        o2 = po.Oracle(sc, seed=3)
        o2.set_instances(sc["blas"], inst)
        assert ctx.trace(rays).tobytes() == o2.trace_closest(rays, brute=True).tobytes()
    assert ctx.stats().trace_overruns == 0
    ctx.close()


def test_refit_of_an_instanced_scene(po):
    sc = scenes.instanced_scene(seed=21, n_instances=10)
    o, ctx = _pair(po, sc, 16, 16)
    v = sc["vertices"].copy()
    rng = np.random.default_rng(3)
    v["p"] += rng.normal(0, 0.02, v["p"].shape).astype(np.float32)
    ctx.update_vertices(v)
    with pytest.raises(capi.PrtError):
        ctx.trace(_rays(10, 1))
    ctx.refit_accel()
    sc_moved = dict(sc, vertices=v)
    o2 = po.Oracle(sc_moved, seed=3)
    o2.set_instances(sc["blas"], sc["instances"])
    rays = _rays(30000, 6)
    assert ctx.trace(rays).tobytes() == o2.trace_closest(rays, brute=True).tobytes()
    ctx.close()


def test_instance_errors():
    sc = scenes.instanced_scene(seed=22, n_instances=3)
    ctx = capi.Context(16, 16)
    inst = sc["instances"].copy()
    with pytest.raises(capi.PrtError):  # no scene yet
        ctx.set_instances(sc["blas"], inst)
    ctx.set_scene(sc)
    bad = inst.copy()
    bad["transform"][1][3] = 0.5  # projective fourth row
    with pytest.raises(capi.PrtError):
        ctx.set_instances(sc["blas"], bad)
    bad = inst.copy()
    bad["blas"][2] = 99
    with pytest.raises(capi.PrtError):
        ctx.set_instances(sc["blas"], bad)
    blas = sc["blas"].copy()
    blas["n_triangles"][0] = 10 ** 9
    with pytest.raises(capi.PrtError):
        ctx.set_instances(blas, inst)
    sing = inst.copy()
    sing["transform"][1] = scenes.affine(scale=(1.0, 0.0, 1.0))
    ctx.set_instances(sc["blas"], sing)
    with pytest.raises(capi.PrtError):  # singular transform: refused at build time
        ctx.build_accel()
    ctx.set_instances(sc["blas"], inst)
    ctx.build_accel()
    ctx.set_instances(sc["blas"][:0], inst[:0])  # back to the single-level scene
    ctx.build_accel()
    assert ctx.read_bvh()[0].shape[0] > 0
    ctx.close()


def test_two_level_context_returns_device_memory():
    """prt_destroy / a rebuild / a return to the single-level scene give back what a two-level scene allocated (trees,
    instance tables): free device memory stays put over create-build-update-render-destroy rounds."""
    import torch

    sc = scenes.instanced_scene(seed=24, n_instances=50, wild=True)
    W, H = 128, 96

    def round_trip(k):
        ctx = capi.Context(W, H)
        ctx.set_scene(sc)
        ctx.set_instances(sc["blas"], sc["instances"])
        ctx.build_accel()
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=2))
        ctx.render(capi.MODE_PATH, 1, 3)
        ctx.update_instance_transforms(sc["instances"]["transform"])
        ctx.build_accel()  # rebuild over the existing structures
        if k % 2:
            ctx.set_instances(sc["blas"][:0], sc["instances"][:0])
            ctx.build_accel()
        ctx.render(capi.MODE_PATH, 1, 2)
        ctx.close()

    for k in range(4):
        round_trip(k)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(4, 12):
        round_trip(k)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), f"device memory shrank by {(free0 - free1) >> 20} MiB over eight contexts"
