"""Refit of the acceleration structure (prt_update_vertices + prt_refit_accel: the `updateOnly` path of
vkRT_BLAS.cpp:209-290 / vkRT_TLAS.cpp:178-279) and resume of a progressive render (prt_write_accum), through the
C-ABI on the GPU. The trace result does not depend on the tree, so after a refit every hit must equal, bit for
bit, the oracle's hit on a tree it built from scratch over the moved vertices."""
import numpy as np
import pytest

from pathtracer_amd import capi, scenes, uniforms

pytestmark = pytest.mark.gpu

MISS = 0xFFFFFFFF


def _moved(sc, seed, amount):
    rng = np.random.default_rng(seed)
    v = sc["vertices"].copy()
    p = v["p"] if v.dtype.names else None
    assert p is not None
    # a smooth deformation plus a little noise: the kind of motion a refit is meant for
    q = p.astype(np.float32)
    q = q + (amount * np.sin(q[:, [1, 2, 0]] * 1.7 + 0.3)).astype(np.float32)
    q = q + rng.uniform(-amount * 0.1, amount * 0.1, q.shape).astype(np.float32)
    v["p"] = q.astype(np.float32)
    out = dict(sc)
    out["vertices"] = v
    return out


def _rays(n, seed, extent):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, capi.RAY_DTYPE)
    rays["o"] = rng.uniform(-extent, extent, (n, 3))
    d = rng.uniform(-extent * 0.7, extent * 0.7, (n, 3)) - rays["o"]
    rays["d"] = d * rng.uniform(0.3, 3.0, (n, 1))
    rays["tmin"], rays["tmax"] = 1e-6, 1.0
    return rays


def _check_tree(nodes, order, sc):
    """Every leaf box is the exact box of its triangles, every inner box the union of its children."""
    v = sc["vertices"]["p"]
    idx = np.asarray(sc["indices"], np.uint32).reshape(-1, 3)
    for i in range(len(nodes) - 1, -1, -1):
        n = nodes[i]
        if n["count"]:
            tri = order[n["first"]:n["first"] + n["count"]]
            pts = v[idx[tri].ravel()]
            assert (n["bmin"] == pts.min(0)).all() and (n["bmax"] == pts.max(0)).all(), i
        else:
            a, b = nodes[n["first"]], nodes[n["first"] + 1]
            assert (n["bmin"] == np.minimum(a["bmin"], b["bmin"])).all(), i
            assert (n["bmax"] == np.maximum(a["bmax"], b["bmax"])).all(), i


@pytest.mark.parametrize("maker,extent,amount", [(lambda: scenes.random_soup(5000), 6.0, 0.3),
                                                 (scenes.cornell_box, 3.0, 0.2),
                                                 (lambda: scenes.conference_room(detail=0.4), 5.0, 0.15)])
def test_refit_equals_rebuild_and_oracle(po, maker, extent, amount):
    sc = maker()
    ctx = capi.Context(16, 16)
    ctx.set_scene(sc)
    ctx.build_accel()
    n0, o0 = ctx.read_bvh()
    rays = _rays(20000, 5, extent)
    for step in range(2):  # two refits in a row: the second starts from refitted boxes
        sc2 = _moved(sc, 10 + step, amount * (step + 1))
        ctx.update_vertices(sc2["vertices"])
        with pytest.raises(capi.PrtError):  # between the two calls the tree is stale
            ctx.trace(rays[:4])
        ctx.refit_accel()
        n1, o1 = ctx.read_bvh()
        assert (o1 == o0).all() and (n1["first"] == n0["first"]).all() and (n1["count"] == n0["count"]).all()
        _check_tree(n1, o1, sc2)
        g = ctx.trace(rays)
        o = po.Oracle(sc2, seed=3)
        o.build_bvh()
        assert g.tobytes() == o.trace_closest(rays).tobytes()
        ga = ctx.trace(rays, any_hit=True)
        assert ((ga["prim"] != MISS) == (o.trace_any(rays) == 1)).all()
        assert ctx.stats().bvh_refit_ms > 0.0
    # a rebuild over the same vertices gives the same hits
    ctx.build_accel()
    assert ctx.trace(rays).tobytes() == g.tobytes()
    ctx.close()


def test_render_after_refit_bit_exact(po):
    sc = scenes.cornell_box()
    sc2 = _moved(sc, 3, 0.05)
    W = H = 96
    ctx = capi.Context(W, H)
    ctx.set_scene(sc)
    ctx.set_seed(3)
    ctx.build_accel()
    ctx.update_vertices(sc2["vertices"])
    ctx.refit_accel()
    o = po.Oracle(sc2, seed=3)
    o.build_bvh()
    acc = np.zeros((H, W, 4), np.float32)
    for it in (1, 2):
        u = uniforms.make_uniforms(sc, W, H, iteration=it)
        acc, ids, rays = o.render(u, W, H, capi.MODE_PATH, accum=acc, want_ids=True)
        ctx.set_uniforms(u)
        ctx.render_iteration(capi.MODE_PATH)
        assert (ctx.read_hit_ids() == ids).all()
    g = ctx.read_accum()
    assert int((g.view(np.uint32) != acc.view(np.uint32)).any(axis=-1).sum()) == 0
    ctx.close()


def test_update_vertices_errors():
    sc = scenes.cornell_box()
    ctx = capi.Context(8, 8)
    with pytest.raises(capi.PrtError):
        ctx.update_vertices(sc["vertices"])  # no scene
    ctx.set_scene(sc)
    with pytest.raises(capi.PrtError):
        ctx.refit_accel()  # nothing built yet
    with pytest.raises(capi.PrtError):
        ctx.update_vertices(sc["vertices"][:-1])  # wrong count
    ctx.update_vertices(sc["vertices"])  # before the first build: just replaces the array
    ctx.build_accel()
    ctx.trace(np.zeros(1, capi.RAY_DTYPE))
    ctx.close()


def test_resume_from_saved_accumulation():
    """Iterations 1..6 in one go == iterations 1..3, image saved, new context, image restored, iterations 4..6."""
    sc = scenes.cornell_box()
    W = H = 64
    u = uniforms.make_uniforms(sc, W, H, iteration=1)

    def fresh():
        c = capi.Context(W, H)
        c.set_scene(sc)
        c.set_seed(9)
        c.build_accel()
        c.set_uniforms(u)
        return c

    a = fresh()
    a.render(capi.MODE_PATH, 1, 6)
    full = a.read_accum()
    a.close()
    b = fresh()
    b.render(capi.MODE_PATH, 1, 3)
    saved = b.read_accum()
    b.close()
    c = fresh()
    c.write_accum(saved)
    c.render(capi.MODE_PATH, 4, 3)
    assert c.read_accum().tobytes() == full.tobytes()
    c.close()
