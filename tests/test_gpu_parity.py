"""GPU parity tests: the HIP path, called through the C-ABI (include/prt.h), against the CPU oracle on the
same seeded inputs. Bar: bit-exact for primitive ids, BVH nodes / ordering and Sobol words; bit-exact for the
fp32 accumulation buffer too, because oracle and kernels evaluate the same fp32 expression trees with
contraction off and share the sin/cos polynomial (tolerance therefore 0 ulp; the north-star bar is only
"rMSE under tolerance" for images). Full-size cases use size-independent properties."""
import numpy as np
import pytest

from pathtracer_amd import capi, scenes, uniforms

pytestmark = pytest.mark.gpu

MISS = 0xFFFFFFFF


def _pair(po, sc, W, H, seed=3, flags=0):
    o = po.Oracle(sc, seed=seed)
    o.build_bvh()
    ctx = capi.Context(W, H)
    ctx.set_scene(sc)
    ctx.set_seed(seed)
    ctx.build_accel()
    ctx.set_flags(flags)
    return o, ctx


def _render_both(po, sc, W, H, mode, iters, flags=0, seed=3, **kw):
    o, ctx = _pair(po, sc, W, H, seed, flags)
    acc = np.zeros((H, W, 4), np.float32)
    for it in iters:
        u = uniforms.make_uniforms(sc, W, H, iteration=it, **kw)
        acc, ids, rays = o.render(u, W, H, mode, accum=acc, flags=flags, want_ids=True)
        ctx.set_uniforms(u)
        ctx.reset_stats()
        ctx.render_iteration(mode)
        g = ctx.read_accum()
        assert (ctx.read_hit_ids() == ids).all(), f"iteration {it}: primary hit ids differ"
        assert ctx.stats().rays == rays, f"iteration {it}: ray count differs"
        nd = int((g.view(np.uint32) != acc.view(np.uint32)).any(axis=-1).sum())
        assert nd == 0, f"iteration {it}: {nd} pixels differ, max abs {np.nanmax(np.abs(g - acc))}"
    ctx.close()


def test_sobol_table_equals_reference_words(po):
    ctx = capi.Context(8, 8)
    assert (ctx.read_sobol() == po.golden_sobol()[:, :32]).all()
    ctx.close()


@pytest.mark.parametrize("maker", [scenes.cornell_box, lambda: scenes.random_soup(5000),
                                   lambda: scenes.random_soup(3), lambda: scenes.conference_room(detail=0.5)])
def test_gpu_bvh_is_the_oracle_bvh(po, maker):
    sc = maker()
    o, ctx = _pair(po, sc, 16, 16)
    gn, go = ctx.read_bvh()
    on, oo = o.bvh()
    assert len(gn) == len(on) and ctx.stats().bvh_depth == o.bvh_depth()
    assert gn.tobytes() == on.tobytes()
    assert (go == oo).all()
    ctx.close()


def test_gpu_bvh_degenerate_centroids(po):
    mb = scenes.MeshBuilder()
    tri = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32)
    mb.add(np.repeat(tri, 37, 0), 0)
    mb.add(tri + 2.0, 0)
    sc = mb.finish([scenes.make_material()])
    o, ctx = _pair(po, sc, 16, 16)
    gn, go = ctx.read_bvh()
    on, oo = o.bvh()
    assert gn.tobytes() == on.tobytes() and (go == oo).all()
    ctx.close()


def _rays(n, seed, extent):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, capi.RAY_DTYPE)
    rays["o"] = rng.uniform(-extent, extent, (n, 3))
    d = rng.uniform(-extent * 0.7, extent * 0.7, (n, 3)) - rays["o"]
    rays["d"] = d * rng.uniform(0.3, 3.0, (n, 1))
    rays["tmin"], rays["tmax"] = 1e-6, 1.0
    return rays


@pytest.mark.parametrize("maker,extent", [(lambda: scenes.random_soup(5000), 6.0), (scenes.cornell_box, 3.0),
                                          (lambda: scenes.conference_room(detail=0.4), 5.0),
                                          (lambda: scenes.random_soup(2), 3.0)])
def test_trace_matches_oracle_and_brute_force(po, maker, extent):
    sc = maker()
    o, ctx = _pair(po, sc, 16, 16)
    rays = _rays(20000, 11, extent)
    g = ctx.trace(rays)
    c = o.trace_closest(rays)
    assert g.tobytes() == c.tobytes()  # prim, u, v, t all bit-exact
    b = o.trace_closest(rays[:4000], brute=True)
    assert g[:4000].tobytes() == b.tobytes()  # and independent of the BVH
    ga = ctx.trace(rays, any_hit=True)
    assert ((ga["prim"] != MISS) == (o.trace_any(rays) == 1)).all()
    ctx.close()


def test_trace_edge_cases(po):
    sc = scenes.cornell_box()
    o, ctx = _pair(po, sc, 16, 16)
    assert len(ctx.trace(np.zeros(0, capi.RAY_DTYPE))) == 0  # empty batch
    rays = np.zeros(6, capi.RAY_DTYPE)
    rays["o"] = (0.1, 1.0, 0.3)
    rays["d"] = [(1, 0, 0), (0, 1, 0), (0, 0, -1), (0, -3, 0), (0, 0, 0), (1e-30, 5, 1e-30)]  # axis aligned, zero
    rays["tmin"] = [1e-6, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6]
    rays["tmax"] = [10, 10, 10, 0.1, 1, 1]
    g, c = ctx.trace(rays), o.trace_closest(rays)
    assert g.tobytes() == c.tobytes()
    assert g["prim"][3] == MISS and g["prim"][4] == MISS
    ctx.close()


def test_nan_rays_are_misses_and_cost_nothing(po):
    """A NaN in the origin or direction (the reference keeps NaN normals for corners without one): every comparison of the
    triangle test fails, so the ray hits nothing — in the oracle after walking the tree, here without walking it."""
    sc = scenes.conference_room(detail=0.4)
    o, ctx = _pair(po, sc, 16, 16)
    rays = _rays(4096, 3, 5.0)
    nan = np.float32(np.nan)
    for k in range(6):
        (rays["o"] if k < 3 else rays["d"])[k::7, k % 3] = nan
    g, c = ctx.trace(rays), o.trace_closest(rays)
    assert g.tobytes() == c.tobytes()
    bad = np.isnan(rays["o"]).any(axis=1) | np.isnan(rays["d"]).any(axis=1)
    assert bad.sum() > 2000 and (g["prim"][bad] == MISS).all() and (g["t"][bad] == rays["tmax"][bad]).all()
    assert (g["prim"][~bad] != MISS).any()
    assert ((ctx.trace(rays, any_hit=True)["prim"] != MISS) == (o.trace_any(rays) == 1)).all()
    assert ctx.stats().trace_overruns == 0
    ctx.close()


def test_grazing_self_hits(po):
    """Rays that leave a thin triangle almost in its plane (tests/helpers/grazing.py): closest and any-hit results of
    the GPU's BVH4 traversal equal the oracle's brute force over all triangles (hit definition)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
    import grazing

    sc = grazing.strip_scene()
    o, ctx = _pair(po, sc, 16, 16)
    for seed in (1, 2, 3):
        rays = grazing.grazing_rays(capi.RAY_DTYPE, sc, 20000, seed)
        g, b = ctx.trace(rays), o.trace_closest(rays, brute=True)
        assert g.tobytes() == b.tobytes()
        assert ((ctx.trace(rays, any_hit=True)["prim"] != MISS) == (o.trace_any(rays, brute=True) == 1)).all()
    ctx.close()


def test_config0_cornell_ao_512(po):
    """BASELINE.json configs[0]: Cornell box 512x512, AO mode, 16 rays."""
    _render_both(po, scenes.cornell_box(), 512, 512, capi.MODE_AO, [1], seed=1, ao_rays=16, ao_length=1.0)


@pytest.mark.parametrize("case", ["cornell", "cornell_literal", "soup_open", "soup_aa4", "soup_aa3_literal", "rr",
                                  "conference", "conference_textured", "bounce0", "ao_multi"])
def test_path_and_ao_bit_exact(po, case):
    if case == "cornell":
        _render_both(po, scenes.cornell_box(), 160, 128, 0, [1, 2, 3], bounces=5, light_area=0.2)
    elif case == "cornell_literal":
        _render_both(po, scenes.cornell_box(), 128, 128, 0, [1, 2, 1, 5], flags=capi.FLAG_LITERAL, bounces=3)
    elif case == "soup_open":
        _render_both(po, scenes.random_soup(4000), 160, 120, 0, [1, 2], bounces=4)
    elif case == "soup_aa4":
        _render_both(po, scenes.random_soup(2000), 96, 64, 0, [1, 5], bounces=2, aa_rays=4)
    elif case == "soup_aa3_literal":
        _render_both(po, scenes.random_soup(2000), 96, 64, 0, [1, 4], flags=capi.FLAG_LITERAL, bounces=2, aa_rays=3)
    elif case == "rr":
        _render_both(po, scenes.cornell_box(), 128, 96, 0, [1, 2, 3], flags=capi.FLAG_RUSSIAN_ROULETTE, bounces=8,
                     light_area=0.3)
    elif case == "conference":
        _render_both(po, scenes.conference_room(detail=0.5), 320, 180, 0, [1, 2], bounces=5)
    elif case == "conference_textured":
        _render_both(po, scenes.conference_room(detail=0.4, textured=True), 320, 180, 0, [1, 2], bounces=5)
    elif case == "bounce0":
        _render_both(po, scenes.cornell_box(), 64, 64, 0, [1], bounces=0)
    elif case == "ao_multi":
        _render_both(po, scenes.random_soup(3000), 128, 96, 1, [0, 2, 3], ao_rays=7, ao_length=2.5)


@pytest.mark.parametrize("W,H,bounces,n,batch", [(1, 1, 3, 5, 4), (7, 3, 0, 3, 8), (67, 45, 10, 9, 4), (130, 1, 2, 17, 8),
                                                 (33, 31, 14, 2, 2)])
def test_odd_sizes_and_extreme_bounces_batched(po, W, H, bounces, n, batch):
    """Image sizes that are no multiple of a wave, a block or a tile, zero bounces, the UI's maximum (10) and the
    library's maximum (14: the shader's scrambleArray[16] ends there) number of bounces, rendered through prt_render (several iterations per pass): image, primary
    hit ids of the last iteration and ray count against the oracle."""
    sc = scenes.cornell_box()
    o, ctx = _pair(po, sc, W, H, 5, 0)
    ctx.set_iteration_batch(batch)
    ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=bounces, light_area=0.3))
    ctx.reset_stats()
    ctx.render(capi.MODE_PATH, 1, n, 1)
    acc, ids, rays = np.zeros((H, W, 4), np.float32), None, 0
    for it in range(1, n + 1):
        acc, ids, r = o.render(uniforms.make_uniforms(sc, W, H, bounces=bounces, light_area=0.3, iteration=it), W, H,
                               capi.MODE_PATH, accum=acc, want_ids=True)
        rays += r
    assert ctx.read_accum().tobytes() == acc.tobytes()
    assert (ctx.read_hit_ids() == ids).all() and ctx.stats().rays == rays and ctx.stats().iterations == n
    with pytest.raises(capi.PrtError):
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=15))
    ctx.close()


def _degenerate_soup():
    """Triangle soup with what real assets contain: zero-area triangles (repeated corner, collinear corners), exact
    duplicates (closest-hit tie -> lower primitive id), a needle thinner than a float ulp at its position, and a copy of
    part of the soup 4000 units away (large coordinates next to small ones)."""
    from pathtracer_amd.scenes import MeshBuilder, make_material

    base = scenes.random_soup(600, seed=11)
    P = base["vertices"]["p"][base["indices"]].reshape(-1, 3, 3).astype(np.float32)
    extra = [P[:40].copy(),  # duplicates
             np.stack([P[40:60, 0], P[40:60, 0], P[40:60, 1]], axis=1),  # repeated corner
             np.stack([P[60:80, 0], (P[60:80, 0] + P[60:80, 1]) * 0.5, P[60:80, 1]], axis=1),  # collinear
             np.array([[[0.0, 0.0, 0.0], [3.0, 0.0, 0.0], [1.5, 1e-9, 0.0]]], np.float32),  # needle
             P[:100] + np.float32(4000.0)]
    mb = MeshBuilder()
    mb.add(np.concatenate([P] + extra), 0)
    sc = mb.finish([make_material()])
    sc["name"], sc["camera"], sc["light"] = "degenerate", base["camera"], base["light"]
    return sc


@pytest.mark.parametrize("case", ["ao_1ray", "ao_256rays", "aa8_b10", "rr_b14", "table_batched", "degenerate", "one_triangle"])
def test_more_edge_cases_through_prt_render(po, case):
    W, H, mode, n, flags, kw, table = 37, 29, capi.MODE_PATH, 5, 0, dict(bounces=3), None
    sc = scenes.cornell_box()
    if case == "ao_1ray":
        mode, n, kw = capi.MODE_AO, 2, dict(ao_rays=1, ao_length=0.001)
    elif case == "ao_256rays":
        mode, n, kw = capi.MODE_AO, 1, dict(ao_rays=256, ao_length=5.0)
    elif case == "aa8_b10":
        n, kw = 3, dict(bounces=10, aa_rays=8, filter_radius=4.0, light_area=0.3)
    elif case == "rr_b14":
        flags, kw = capi.FLAG_RUSSIAN_ROULETTE, dict(bounces=14, light_area=0.3)
    elif case == "table_batched":
        table = np.random.default_rng(5).integers(0, 2**32, (32, H * W), dtype=np.uint64).astype(np.uint32)
        n = 7
    elif case == "degenerate":
        sc, W, H, n = _degenerate_soup(), 96, 64, 3
    elif case == "one_triangle":
        sc, W, H = scenes.random_soup(1, seed=2, extent=0.1, size=2.0), 64, 48
    o, ctx = _pair(po, sc, W, H, 4, flags)
    if table is not None:
        ctx.upload_scrambles(table)
        o.set_scramble_table(table, W, H)
    ctx.set_iteration_batch(4)
    ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, **kw))
    ctx.reset_stats()
    ctx.render(mode, 1, n, 1)
    acc, ids, rays = np.zeros((H, W, 4), np.float32), None, 0
    for it in range(1, n + 1):
        acc, ids, r = o.render(uniforms.make_uniforms(sc, W, H, iteration=it, **kw), W, H, mode, accum=acc, flags=flags,
                               want_ids=True)
        rays += r
    g = ctx.read_accum()
    nd = int((g.view(np.uint32) != acc.view(np.uint32)).any(axis=-1).sum())
    assert nd == 0, f"{nd} pixels differ"
    assert (ctx.read_hit_ids() == ids).all() and ctx.stats().rays == rays
    ctx.close()


def test_hit_ids_after_a_batched_render_then_ao(po):
    """prt_read_hit_ids returns the ids of the LAST iteration: after a batched path render followed by an AO iteration
    those are the AO pass's primary ids (pixel-centre rays), not the stale ids of the batch."""
    sc = scenes.cornell_box()
    W, H = 64, 48
    o, ctx = _pair(po, sc, W, H)
    ctx.set_iteration_batch(4)
    ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=2, iteration=1))
    ctx.render(capi.MODE_PATH, 1, 8)
    u = uniforms.make_uniforms(sc, W, H, ao_rays=4, iteration=1)
    ctx.set_uniforms(u)
    ctx.clear_accum()
    ctx.render_iteration(capi.MODE_AO)
    acc, ids, _ = o.render(u, W, H, capi.MODE_AO, want_ids=True)
    assert (ctx.read_hit_ids() == ids).all() and ctx.read_accum().tobytes() == acc.tobytes()
    assert ctx.stats().trace_overruns == 0
    ctx.close()


def test_pipelined_iterations_equal_sequential():
    """Iterations in flight on several streams (prt_set_pipeline_depth) sum in iteration order: bit-identical."""
    sc = scenes.conference_room(detail=0.4)
    W, H = 320, 180
    out = []
    for depth in (1, 4, 8):
        ctx = capi.Context(W, H)
        ctx.set_scene(sc)
        ctx.set_seed(9)
        ctx.build_accel()
        ctx.set_pipeline_depth(depth)
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=5))
        ctx.render(capi.MODE_PATH, 1, 11, 1)
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=2, iteration=12))  # fewer terms per pass
        ctx.render_iteration(capi.MODE_PATH)
        out.append((ctx.read_accum().tobytes(), ctx.stats().rays, ctx.read_hit_ids().tobytes()))
        ctx.close()
    assert out[0] == out[1] == out[2]
    with pytest.raises(capi.PrtError):
        capi.Context(8, 8).set_pipeline_depth(99)


def test_batched_ao_iterations_equal_one_by_one(po):
    """AO mode batches like path mode (prt_render): several iterations in one pass, their terms added to the image in
    iteration order. Image, ray count, iteration count and hit ids equal one iteration per pass and the oracle's;
    a camera that sees past the scene covers the miss term; a path render afterwards still works on the grown buffers;
    the literal image semantics are not batched."""
    sc = scenes.random_soup(600, seed=4, extent=3.0, size=0.6)
    W, H = 160, 96
    table = np.random.default_rng(5).integers(0, 2**32, (32, H, W), dtype=np.uint64).astype(np.uint32)
    kw = dict(ao_rays=7, ao_length=0.8)
    out = []
    for batch in (1, 2, 5, 64):
        ctx = capi.Context(W, H)
        ctx.set_scene(sc)
        ctx.set_seed(11)
        ctx.build_accel()
        ctx.set_iteration_batch(batch)
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, **kw))
        ctx.render(capi.MODE_AO, 1, 11, 1)
        ids = ctx.read_hit_ids().tobytes()
        ctx.upload_scrambles(table)
        ctx.render(capi.MODE_AO, 12, 3, 2)
        img_ao = ctx.read_accum().tobytes()
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=2, **kw))
        ctx.render(capi.MODE_PATH, 1, 3, 1)
        out.append((img_ao, ctx.read_accum().tobytes(), ctx.stats().rays, ctx.stats().iterations, ids))
        ctx.close()
    assert all(o == out[0] for o in out[1:])
    assert out[0][3] == 17
    o = po.Oracle(sc, seed=11)
    o.build_bvh()
    acc, last, rays = np.zeros((H, W, 4), np.float32), None, 0
    for it in range(1, 12):
        acc, last, r = o.render(uniforms.make_uniforms(sc, W, H, iteration=it, **kw), W, H, capi.MODE_AO, accum=acc, want_ids=True)
        rays += r
    o.set_scramble_table(table.reshape(32, H * W), W, H)
    for it in (12, 14, 16):
        acc, _, r = o.render(uniforms.make_uniforms(sc, W, H, iteration=it, **kw), W, H, capi.MODE_AO, accum=acc, want_ids=True)
        rays += r
    assert out[0][0] == acc.tobytes() and out[0][4] == last.tobytes()
    assert (acc[..., 3] == 0).any() and (acc[..., 3] > 0).any()  # misses and hits
    # literal semantics: one iteration per pass, whatever the batch says (the image restarts at iteration <= 1)
    res = []
    for batch in (1, 8):
        ctx = capi.Context(W, H)
        ctx.set_scene(sc)
        ctx.set_seed(11)
        ctx.build_accel()
        ctx.set_flags(capi.FLAG_LITERAL)
        ctx.set_iteration_batch(batch)
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, **kw))
        ctx.render(capi.MODE_AO, 0, 5, 1)
        res.append(ctx.read_accum().tobytes())
        ctx.close()
    assert res[0] == res[1]


def test_batched_iterations_equal_one_by_one(po):
    """prt_render puts several iterations into one pass over n*W*H pixels (prt_set_iteration_batch): the image,
    the ray count and the hit ids of the last iteration equal those of one iteration per pass — also with a stride
    (multi-GPU share), Russian roulette, an uploaded scramble table, and a batch that does not divide the count."""
    sc = scenes.conference_room(detail=0.4)
    W, H = 200, 120
    table = np.random.default_rng(3).integers(0, 2**32, (32, H, W), dtype=np.uint64).astype(np.uint32)
    out = []
    for batch, depth in ((1, 1), (1, 3), (2, 2), (5, 3), (8, 2), (4, 1)):
        ctx = capi.Context(W, H)
        ctx.set_scene(sc)
        ctx.set_seed(9)
        ctx.build_accel()
        ctx.set_pipeline_depth(depth)
        ctx.set_iteration_batch(batch)
        assert ctx.iteration_batch() == batch
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=4))
        ctx.render(capi.MODE_PATH, 2, 13, 3)  # iterations 2, 5, ..., 38
        ids = ctx.read_hit_ids().tobytes()
        ctx.set_flags(capi.FLAG_RUSSIAN_ROULETTE)
        ctx.upload_scrambles(table)
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=6))  # more terms per pass than before
        ctx.render(capi.MODE_PATH, 40, 7, 1)
        out.append((ctx.read_accum().tobytes(), ctx.stats().rays, ctx.stats().iterations, ids, ctx.read_hit_ids().tobytes()))
        ctx.close()
    assert all(o == out[0] for o in out[1:])
    assert out[0][2] == 20
    # and against the oracle for the first part
    o = po.Oracle(sc, seed=9)
    o.build_bvh()
    acc, last = np.zeros((H, W, 4), np.float32), None
    ctx = capi.Context(W, H)
    ctx.set_scene(sc)
    ctx.set_seed(9)
    ctx.build_accel()
    ctx.set_iteration_batch(4)
    ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=2))
    ctx.render(capi.MODE_PATH, 1, 6, 1)
    for it in range(1, 7):
        acc, last, _ = o.render(uniforms.make_uniforms(sc, W, H, bounces=2, iteration=it), W, H, capi.MODE_PATH, accum=acc,
                                want_ids=True)
    assert ctx.read_accum().tobytes() == acc.tobytes() and (ctx.read_hit_ids() == last).all()
    with pytest.raises(capi.PrtError):
        ctx.set_iteration_batch(0)
    ctx.close()


def test_uploaded_scramble_table(po):
    """The reference's explicit 32-layer scramble image (vkRTX_setup.cpp:808-911) instead of the hash."""
    sc = scenes.cornell_box()
    W, H = 64, 48
    table = np.random.default_rng(123).integers(0, 2 ** 32, (32, H * W), dtype=np.uint64).astype(np.uint32)
    o, ctx = _pair(po, sc, W, H)
    o.set_scramble_table(table, W, H)
    ctx.upload_scrambles(table)
    u = uniforms.make_uniforms(sc, W, H, bounces=3, iteration=1)
    ctx.set_uniforms(u)
    ctx.render_iteration(0)
    acc, _, _ = o.render(u, W, H, 0)
    assert ctx.read_accum().tobytes() == acc.tobytes()
    ctx.close()


def test_context_lifecycle_returns_device_memory():
    """prt_destroy gives back everything prt_create / prt_render allocated (pipes grown by batched passes included):
    free device memory stays put over ten create-render-destroy rounds (after the runtime's own pools have formed)."""
    import torch

    sc = scenes.cornell_box()
    W, H = 256, 192

    def round_trip(k):
        ctx = capi.Context(W, H)
        ctx.set_scene(sc)
        ctx.build_accel()
        ctx.set_pipeline_depth(1 + k % 3)
        ctx.set_iteration_batch(1 + (3 * k) % 8)
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=2 + k % 3))
        ctx.render(capi.MODE_PATH, 1, 9, 1)
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, ao_rays=4))
        ctx.render_iteration(capi.MODE_AO)
        img = ctx.read_accum()
        ctx.close()
        return img

    first = round_trip(0)
    for k in range(1, 4):  # the HIP runtime keeps per-stream pools it creates in the first rounds (a few hundred MiB)
        round_trip(k)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(4, 14):
        round_trip(k)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), f"device memory shrank by {(free0 - free1) >> 20} MiB over ten contexts"
    assert np.isfinite(first[..., 3]).all()


def test_profiling_mask_restricts_the_brackets():
    sc = scenes.cornell_box()
    ctx = capi.Context(128, 96)
    ctx.set_scene(sc)
    ctx.build_accel()
    ctx.set_uniforms(uniforms.make_uniforms(sc, 128, 96, bounces=2))
    ctx.set_profiling(1)
    ctx.set_profiling_mask(1 << capi.K_SHADE)
    ctx.reset_stats()
    ctx.render(capi.MODE_PATH, 1, 4, 1)
    p = ctx.profile()
    assert p.launches[capi.K_SHADE] > 0 and p.ms[capi.K_SHADE] > 0.0
    assert all(p.launches[k] == 0 for k in range(capi.K_COUNT) if k != capi.K_SHADE)
    ctx.set_profiling_mask(0xffffffff)
    ctx.reset_stats()
    ctx.render(capi.MODE_PATH, 5, 4, 1)
    p = ctx.profile()
    assert p.launches[capi.K_GENERATE] > 0 and p.launches[capi.K_RESOLVE] > 0
    assert p.fused_closest_rays + p.fused_any_rays <= p.closest_rays + p.any_rays
    ctx.close()


def test_api_errors():
    ctx = capi.Context(32, 32)
    with pytest.raises(capi.PrtError):
        ctx.render_iteration(0)  # no scene
    with pytest.raises(capi.PrtError):
        ctx.build_accel()
    sc = scenes.cornell_box()
    bad = dict(sc)
    bad["indices"] = sc["indices"].copy()
    bad["indices"][5] = 10 ** 6
    with pytest.raises(capi.PrtError):
        ctx.set_scene(bad)
    ctx.set_scene(sc)
    ctx.build_accel()
    with pytest.raises(capi.PrtError):
        ctx.render_iteration(7)
    u = uniforms.make_uniforms(sc, 32, 32, bounces=99)
    with pytest.raises(capi.PrtError):
        ctx.set_uniforms(u)
    # ragged index count: trailing indices that do not form a triangle are ignored
    rag = dict(sc)
    rag["indices"] = np.concatenate([sc["indices"], sc["indices"][:2]])
    ctx.set_scene(rag)
    ctx.build_accel()
    assert ctx.stats().bvh_nodes > 0
    ctx.close()


def test_postprocess_matches_oracle(po):
    sc = scenes.cornell_box()
    W = H = 64
    o, ctx = _pair(po, sc, W, H)
    u = uniforms.make_uniforms(sc, W, H, ao_rays=8)
    ctx.set_uniforms(u)
    ctx.render_iteration(1)
    img = ctx.postprocess()
    ref = po.postprocess(ctx.read_accum(), W, H)
    assert img.tobytes() == ref.tobytes()  # byte output: exact (oracle and kernel share one definition of pow(x, 1/2.2))
    # a path-traced image (radiance well above 1, zeros), and hand-made extremes: NaN, infinities, negative, w == 0
    ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=3, iteration=1))
    ctx.clear_accum()
    ctx.render(0, 1, 3)
    assert ctx.postprocess().tobytes() == po.postprocess(ctx.read_accum(), W, H).tobytes()
    rng = np.random.default_rng(5)
    acc = rng.uniform(0, 4, (H, W, 4)).astype(np.float32) * rng.choice([1e-30, 1e-3, 1, 1e3, 1e30], (H, W, 1)).astype(np.float32)
    acc[0, :8, 0] = [np.nan, np.inf, -np.inf, -1.0, 0.0, -0.0, 1e-45, 3.4e38]
    acc[1, :8, 3] = [0.0, np.nan, np.inf, -1.0, 1e-45, 3.4e38, 1.0, 2.0]
    ctx.write_accum(acc)
    assert ctx.postprocess().tobytes() == po.postprocess(acc, W, H).tobytes()
    ctx.close()


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[1] at full size: properties that do not need a full oracle render
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def conference_full():
    sc = scenes.conference_room(seed=2, detail=1.4)
    ctx = capi.Context(1920, 1080)
    ctx.set_scene(sc)
    ctx.set_seed(2)
    ctx.build_accel()
    yield sc, ctx
    ctx.close()


def test_full_size_primary_hit_ids_bit_exact(po, conference_full):
    sc, ctx = conference_full
    W, H = 1920, 1080
    u = uniforms.make_uniforms(sc, W, H, bounces=0, iteration=1)
    ctx.clear_accum()
    ctx.set_uniforms(u)
    ctx.render_iteration(0)
    gid = ctx.read_hit_ids()
    o = po.Oracle(sc, seed=2)
    _, ids, _ = o.render(u, W, H, 0, want_ids=True)
    assert (gid == ids).all()
    assert (gid != MISS).all()  # closed room


def test_full_size_image_bit_exact(po, conference_full):
    """BASELINE.json's metric configuration itself — 328 k triangles, 1920x1080, 5 bounces — for three iterations
    through the batched prt_render: the whole accumulation image and the ray count against the oracle."""
    sc, ctx = conference_full
    W, H, B = 1920, 1080, 5
    ctx.clear_accum()
    ctx.reset_stats()
    ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=B))
    ctx.render(capi.MODE_PATH, 1, 3, 1)
    g = ctx.read_accum()
    o = po.Oracle(sc, seed=2)
    acc, rays = np.zeros((H, W, 4), np.float32), 0
    for it in (1, 2, 3):
        acc, _, r = o.render(uniforms.make_uniforms(sc, W, H, bounces=B, iteration=it), W, H, 0, accum=acc)
        rays += r
    nd = int((g.view(np.uint32) != acc.view(np.uint32)).any(axis=-1).sum())
    assert nd == 0 and ctx.stats().rays == rays, f"{nd} of {W * H} pixels differ, rays {ctx.stats().rays} vs {rays}"


def test_full_size_properties(conference_full):
    sc, ctx = conference_full
    W, H, B = 1920, 1080, 5

    def run(iters):
        ctx.clear_accum()
        ctx.reset_stats()
        for it in iters:
            ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=B, iteration=it))
            ctx.render_iteration(0)
        return ctx.read_accum(), ctx.stats().rays

    a, rays = run([1, 2, 3, 4])
    b, rays_b = run([1, 2, 3, 4])
    assert a.tobytes() == b.tobytes() and rays == rays_b  # deterministic
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all()
    w = a[..., 3]
    assert (w == np.round(w)).all() and w.min() >= 4 and w.max() <= 4 * (B + 1)  # vertices per path in [1, B+1]
    assert W * H * 4 * 2 <= rays <= W * H * 4 * (2 * B + 2)  # <= 2B+2 rays per sample
    # sample sharding (multi-GPU rule): shards sum to the whole within fp32 reassociation
    s0, r0 = run([1, 3])
    s1, r1 = run([2, 4])
    assert r0 + r1 == rays
    assert (s0[..., 3] + s1[..., 3] == w).all()
    num = ((s0 + s1 - a)[..., :3] ** 2).mean()
    den = (a[..., :3] ** 2).mean()
    assert num / den < 1e-10


@pytest.mark.gpu
def test_render_degrades_when_a_pass_does_not_fit():
    """Memory policy of prt_render: a pass whose buffers cannot be had (here: more than prt_set_pass_memory_limit allows —
    the same path a hipErrorOutOfMemory takes) is retried with half the iterations, down to one per pass; the image and
    the ray count are those of the unlimited render, and prt_get_pass_iterations reports the batch that was used.
    AO, whose one-iteration form shares pipe 0's buffers, keeps working after a pass that did not fit."""
    sc = scenes.conference_room(detail=0.3)
    W, H = 160, 96
    per_px = 218 + 16 * 3  # bytes per pixel and iteration at 3 bounces (prt.h)
    ref = None
    for limit, want in ((0, 8), (W * H * per_px * 5, 4), (W * H * per_px * 2 + 64, 2), (1, 1)):
        with capi.Context(W, H) as ctx:
            ctx.set_scene(sc)
            ctx.set_seed(4)
            ctx.build_accel()
            ctx.set_iteration_batch(8)
            ctx.set_pass_memory_limit(limit)
            ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=3))
            assert ctx.pass_iterations() == 0
            ctx.render(capi.MODE_PATH, 1, 16, 1)
            assert ctx.pass_iterations() == want
            got = (ctx.read_accum().tobytes(), ctx.stats().rays, ctx.stats().iterations)
            if ref is None:
                ref = got
            assert got == ref
            # AO after a refused pass: batched (refused again, then smaller) and one iteration per call
            ctx.clear_accum()
            ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, ao_rays=4))
            ctx.render(capi.MODE_AO, 1, 6, 1)
            ctx.render_iteration(capi.MODE_AO)
            ao = ctx.read_accum()
            assert np.isfinite(ao).all() and (ao[..., 3] == 7.0).all()
