"""The whole drop-in path on the GPU: OBJ/MTL/textures -> Model -> Renderer (C++ host layer) -> prt_* C-ABI -> HIP
kernels, checked bit for bit against the CPU oracle fed with the same arrays and the same 464-byte block."""
import os

import numpy as np
import pytest

from pathtracer_amd import hostapi

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("mode,light", [(0, None), (0, ((0.0, 1.9, 0.0), (0.0, -1.0, 0.0))), (1, None)])
def test_renderer_from_obj_equals_oracle(po, mode, light):
    obj = os.path.join(G, "obj", "mixed.obj")
    W, H = 96, 64
    s = hostapi.default_settings(numIndicesBounces=4, rtRenderingMode=mode, lightSourceArea=0.25, numAOrays=5, aoRayLength=1.5)
    eye, target = (0.2, 1.1, 0.9), (0.0, 0.6, -0.4)
    first, n = (3, 4) if mode == 0 else (0, 1)
    gpu, ubo = hostapi.render_obj(obj, W, H, s, eye, target, light=light, first=first, n=n)
    scene = hostapi.load_model(obj)
    assert len(scene["textures"]) == 5 and scene["textures"][0].shape == (16, 24, 4)
    o = po.Oracle(scene, seed=1)
    acc = np.zeros((H, W, 4), np.float32)
    for it in range(first, first + n):
        u = ubo.copy()
        u["iteration"] = it
        acc, _, _ = o.render(u, W, H, mode, accum=acc)
    # mixed.obj has faces without normals in a file that has normals: their corners keep n = 0 (Model.cpp:171-175),
    # normalize(0) is NaN in the shader, in the oracle and here alike; NaNs are compared by position
    assert gpu[..., 3].max() > 0 and np.isfinite(gpu).mean() > 0.5
    nan = np.isnan(gpu)
    assert (nan == np.isnan(acc)).all()
    assert np.where(nan, 0, gpu).tobytes() == np.where(nan, 0, acc).tobytes()


def test_renderer_errors_surface_as_messages(tmp_path):
    s = hostapi.default_settings()
    with pytest.raises(hostapi.HostError):
        hostapi.render_obj(str(tmp_path / "none.obj"), 16, 16, s, (0, 0, 1), (0, 0, 0))
    empty = tmp_path / "empty.obj"
    empty.write_text("v 0 0 0\n")
    with pytest.raises(hostapi.HostError):  # no triangles: prt_set_scene refuses, Renderer throws (VK_CHECK_RESULT style)
        hostapi.render_obj(str(empty), 16, 16, s, (0, 0, 1), (0, 0, 0))


def _view_ubo(W, H, s, eye, target, light):
    # the parameter block depends on camera / light / settings only: take it from a render of the plain model
    _, ubo = hostapi.render_obj(os.path.join(G, "obj", "mixed.obj"), W, H, s, eye, target, light=light, first=1, n=1)
    return ubo


def test_instanced_scene_equals_oracle_on_flattened_arrays(po, tmp_path):
    """Two models, four instances (Scene.h) -> one flattened model -> Renderer; oracle fed with the same arrays."""
    (tmp_path / "quad.mtl").write_text("newmtl floor\nKd 0.7 0.7 0.6\nPr 0.4\n")
    (tmp_path / "quad.obj").write_text("mtllib quad.mtl\nv -4 0 -4\nv 4 0 -4\nv 4 0 4\nv -4 0 4\nvn 0 1 0\nusemtl floor\n"
                                       "f 1//1 3//1 2//1\nf 1//1 4//1 3//1\n")
    a = hostapi.open_model(os.path.join(G, "obj", "mixed.obj"))
    q = hostapi.open_model(str(tmp_path / "quad.obj"))
    ident = np.eye(4, dtype=np.float32)

    def xf(tx, ty, tz, sc, rot=0.0):
        c, s_ = np.float32(np.cos(rot)), np.float32(np.sin(rot))
        M = np.zeros((4, 4), np.float32)
        M[0] = (c * sc, 0, -s_ * sc, 0)
        M[1] = (0, sc, 0, 0)
        M[2] = (s_ * sc, 0, c * sc, 0)
        M[3] = (tx, ty, tz, 1)
        return M

    flat, first = hostapi.flatten_instances([a, q], [(1, xf(0, -0.05, 0, 1.0), 0), (0, ident, 1),
                                                     (0, xf(1.6, 0, -0.8, 0.6, 0.7), 2), (0, xf(-1.5, 0.1, -1.0, 0.8, -1.1), 3)])
    scene = flat.scene()
    assert first[-1] == len(scene["indices"]) // 3
    W, H = 96, 64
    s = hostapi.default_settings(numIndicesBounces=3, rtRenderingMode=0, lightSourceArea=0.3)
    eye, target = (0.3, 1.6, 2.6), (0.0, 0.5, -0.5)
    light = ((0.0, 3.0, 0.5), (0.0, -1.0, -0.1))
    gpu = hostapi.render_model(flat, W, H, s, eye, target, light=light, first=1, n=3)
    ubo = _view_ubo(W, H, s, eye, target, light)
    o = po.Oracle(scene, seed=1)
    acc = np.zeros((H, W, 4), np.float32)
    ids = None
    for it in (1, 2, 3):
        u = ubo.copy()
        u["iteration"] = it
        acc, ids, _ = o.render(u, W, H, 0, accum=acc, want_ids=True)
    hit = ids[ids != 0xFFFFFFFF]
    inst = np.searchsorted(first, hit, side="right") - 1
    assert set(np.unique(inst).tolist()) == {0, 1, 2, 3}  # every instance is in view
    nan = np.isnan(gpu)
    assert (nan == np.isnan(acc)).all()
    assert np.where(nan, 0, gpu).tobytes() == np.where(nan, 0, acc).tobytes()
    for m in (a, q, flat):
        m.close()


def test_device_instancing_through_the_host_layer(po, tmp_path):
    """Two models, four instances -> Renderer::initRaytracing(models, instances): one geometry pool + prt_set_instances
    (two-level traversal on the GPU). Bit-exact against the oracle fed with the same pool and instance list; within
    rounding of the flattened path; Renderer::updateInstances = a fresh build with the new transforms."""
    (tmp_path / "quad.mtl").write_text("newmtl floor\nKd 0.7 0.7 0.6\nPr 0.4\n")
    (tmp_path / "quad.obj").write_text("mtllib quad.mtl\nv -4 0 -4\nv 4 0 -4\nv 4 0 4\nv -4 0 4\nvn 0 1 0\nusemtl floor\n"
                                       "f 1//1 3//1 2//1\nf 1//1 4//1 3//1\n")
    a = hostapi.open_model(os.path.join(G, "obj", "mixed.obj"))
    q = hostapi.open_model(str(tmp_path / "quad.obj"))
    ident = np.eye(4, dtype=np.float32)

    def xf(tx, ty, tz, sc, rot=0.0):
        c, s_ = np.float32(np.cos(rot)), np.float32(np.sin(rot))
        M = np.zeros((4, 4), np.float32)
        M[0] = (c * sc, 0, -s_ * sc, 0)
        M[1] = (0, sc, 0, 0)
        M[2] = (s_ * sc, 0, c * sc, 0)
        M[3] = (tx, ty, tz, 1)
        return M

    insts = [(1, xf(0, -0.05, 0, 1.0), 0), (0, ident, 1), (0, xf(1.6, 0, -0.8, 0.6, 0.7), 2),
             (0, xf(-1.5, 0.1, -1.0, 0.8, -1.1), 3)]
    moved = [(1, xf(0, -0.05, 0, 1.0), 0), (0, xf(0.2, 0, 0.1, 1.1, 0.3), 1), (0, xf(1.2, 0.2, -0.5, 0.7, 1.9), 2),
             (0, xf(-1.1, 0.0, -1.3, 0.9, 0.4), 3)]
    W, H = 96, 64
    s = hostapi.default_settings(numIndicesBounces=3, rtRenderingMode=0, lightSourceArea=0.3)
    eye, target = (0.3, 1.6, 2.6), (0.0, 0.5, -0.5)
    light = ((0.0, 3.0, 0.5), (0.0, -1.0, -0.1))
    ubo = _view_ubo(W, H, s, eye, target, light)
    pool, ranges, pinst = hostapi.pool_instances([a, q], insts)
    scene = pool.scene()
    assert int(ranges["n_triangles"].sum()) == len(scene["indices"]) // 3 and (pinst["blas"] == [1, 0, 0, 0]).all()

    def oracle_image(instances):
        o = po.Oracle(scene, seed=1)
        o.set_instances(ranges, instances)
        acc = np.zeros((H, W, 4), np.float32)
        for it in (1, 2, 3):
            u = ubo.copy()
            u["iteration"] = it
            acc, ids, _ = o.render(u, W, H, 0, accum=acc, want_ids=True)
        return acc, ids

    def same(x, y):
        nan = np.isnan(x)
        return (nan == np.isnan(y)).all() and np.where(nan, 0, x).tobytes() == np.where(nan, 0, y).tobytes()

    gpu, gids = hostapi.render_instances([a, q], insts, W, H, s, eye, target, light=light, first=1, n=3)
    acc, ids = oracle_image(pinst)
    assert (gids == ids).all() and same(gpu, acc)
    flat, fids = hostapi.render_instances([a, q], insts, W, H, s, eye, target, light=light, first=1, n=3, flatten=True)
    assert (gids == fids).mean() > 0.999
    ok = ~(np.isnan(gpu) | np.isnan(flat))
    # a last-bit difference in (u, v, t) can send a specular path elsewhere: a few pixels differ visibly, the rest in rounding
    assert float(((gpu[ok] - flat[ok]) ** 2).sum() / (flat[ok] ** 2).sum()) < 1e-3
    assert (np.abs(gpu - flat)[ok] <= 1e-4 * (1.0 + np.abs(flat[ok]))).mean() > 0.99
    # the top level's updateOnly path
    upd, uids = hostapi.render_instances([a, q], insts, W, H, s, eye, target, light=light, first=1, n=3, moved=moved)
    _, _, minst = hostapi.pool_instances([a, q], moved)
    macc, mids = oracle_image(minst)
    assert (uids == mids).all() and same(upd, macc) and not same(upd, gpu)
    for m in (a, q, pool):
        m.close()


def test_checkpoint_resume_is_bit_exact(tmp_path):
    m = hostapi.open_model(os.path.join(G, "obj", "mixed.obj"))
    W, H = 64, 48
    s = hostapi.default_settings(numIndicesBounces=3)
    eye, target = (0.2, 1.1, 0.9), (0.0, 0.6, -0.4)
    full = hostapi.render_model(m, W, H, s, eye, target, first=1, n=6)
    ck = str(tmp_path / "half.ckpt")
    half = hostapi.render_model(m, W, H, s, eye, target, first=1, n=3, save_to=ck)
    c = hostapi.checkpoint_read(ck)
    assert c["next_iteration"] == 4 and c["bounces"] == 3
    assert np.where(np.isnan(half), 0, half).tobytes() == np.where(np.isnan(c["accum"]), 0, c["accum"]).tobytes()
    rest = hostapi.render_model(m, W, H, s, eye, target, first=99, n=3, resume_from=ck)  # `first` comes from the file
    assert (np.isnan(rest) == np.isnan(full)).all()
    assert np.where(np.isnan(rest), 0, rest).tobytes() == np.where(np.isnan(full), 0, full).tobytes()
    with pytest.raises(hostapi.HostError):  # other image size
        hostapi.render_model(m, W, H + 1, s, eye, target, n=1, resume_from=ck)
    s2 = hostapi.default_settings(numIndicesBounces=4)
    with pytest.raises(hostapi.HostError):  # other bounce count
        hostapi.render_model(m, W, H, s2, eye, target, n=1, resume_from=ck)
    m.close()


def test_cli_resume_and_png(tmp_path):
    """prt_render: 4 iterations in one run == 2 + 2 with --checkpoint / --resume; --png decodes to the --out pixels."""
    import subprocess
    exe = os.path.join(os.path.dirname(hostapi.LIB_PATH), "prt_render")
    obj = os.path.join(G, "obj", "mixed.obj")
    common = [exe, obj, "--width", "48", "--height", "32", "--bounces", "2", "--eye", "0.2", "1.1", "0.9", "--target", "0",
              "0.6", "-0.4"]
    run = lambda *a: subprocess.run(common + list(a), check=True, capture_output=True, timeout=120)
    t = lambda n: str(tmp_path / n)
    run("--spp", "4", "--pfm", t("a.pfm"), "--out", t("a.ppm"), "--png", t("a.png"))
    run("--spp", "2", "--checkpoint", t("s.ckpt"), "--out", t("b0.ppm"))
    run("--spp", "2", "--resume", t("s.ckpt"), "--pfm", t("b.pfm"), "--out", t("b.ppm"))
    assert open(t("a.pfm"), "rb").read() == open(t("b.pfm"), "rb").read()
    ppm = open(t("a.ppm"), "rb").read()
    px = np.frombuffer(ppm[ppm.index(b"255\n") + 4:], np.uint8).reshape(32, 48, 3)
    png = hostapi.load_image(t("a.png"))
    assert (png[..., :3] == px).all()


# ---- the frame loop: vkContext::renderFrame / updateGraphicsUniforms (vkContext.cpp:124-231, 993-1060) ---------------------
def _expected_frames(spp, aa, ao_rays, n_frames, events):
    """The bookkeeping of vkContext.cpp:1028-1048 and :164, restated: per frame (ubo.iteration, iteration before, after,
    traced, light captured before this frame's block was filled)."""
    it, mode, camera_moved, move_light, light_is_camera = 1, 0, True, True, False
    out = []
    for f in range(n_frames):
        for (ef, kind, v) in events:
            if ef != f:
                continue
            if kind == hostapi.EVENT_MOVE_CAMERA:
                camera_moved = True
            elif kind == hostapi.EVENT_MOVE_LIGHT:
                move_light = True
            else:
                mode = int(v[0])
        before, ubo_it = it, 0  # UniformBufferObject's default: AO mode never writes it (vkContext.h:79)
        if mode == 0 and it < spp:
            ubo_it = it
            it += aa
        if mode == 1 and it < ao_rays:
            it += 1
        light_in_block = light_is_camera  # the block gets LAST frame's light; m_moveLight is served afterwards
        if move_light:
            light_is_camera, move_light = True, False
        if camera_moved:
            camera_moved, it = False, 1
        out.append(dict(ubo_iteration=ubo_it, before=before, after=it, traced=it < spp, light_is_camera=light_in_block, mode=mode))
    return out


@pytest.mark.parametrize("aa", [1, 2])
def test_frame_loop_follows_the_reference_bookkeeping(po, aa):
    obj = os.path.join(G, "obj", "mixed.obj")
    W, H, spp = 48, 32, 7
    s = hostapi.default_settings(numIndicesBounces=2, samplesPerPixel=spp, numAArays=aa, lightSourceArea=0.3, numAOrays=4,
                                 aoRayLength=1.2)
    eye, target = (0.2, 1.1, 0.9), (0.0, 0.6, -0.4)
    events = [(4, hostapi.EVENT_MOVE_CAMERA, (0, 0, 1)), (6, hostapi.EVENT_MOVE_LIGHT, (0, 0, 0)),
              (13, hostapi.EVENT_SET_MODE, (1, 0, 0)), (13, hostapi.EVENT_MOVE_CAMERA, (1, 0, 0)),
              (17, hostapi.EVENT_SET_MODE, (0, 0, 0)), (17, hostapi.EVENT_MOVE_CAMERA, (-1, 0, 0))]
    n_frames = 24
    with hostapi.open_model(obj) as m:
        rec, gpu = hostapi.frame_loop(m, W, H, s, eye, target, events, n_frames)
    exp = _expected_frames(spp, aa, 4, n_frames, events)
    for f, (r, e) in enumerate(zip(rec, exp)):
        assert (int(r["ubo"]["iteration"]), r["before"], r["after"], r["traced"]) == \
               (e["ubo_iteration"], e["before"], e["after"], e["traced"]), f
        # the light: identity until the first capture has been served, then inverse(view) of the frame that served it
        inv_view = np.linalg.inv(r["ubo"]["view"].reshape(4, 4).T.astype(np.float64)).T.reshape(16)
        if not e["light_is_camera"]:
            assert r["ubo"]["lightTransform"].tobytes() == np.eye(4, dtype=np.float32).tobytes(), f
    # iteration 1 twice after every (re)start, the counter stops one short of spp (vkContext.cpp:164), AO stays at 0
    its = [int(r["ubo"]["iteration"]) for r in rec]
    assert its[:3] == [1, 1, 1 + aa] and all(its[f] == 0 for f in range(13, 17))
    assert rec[1]["ubo"]["lightTransform"].tobytes() != np.eye(4, dtype=np.float32).tobytes()  # one frame late
    assert rec[7]["ubo"]["lightTransform"].tobytes() != rec[6]["ubo"]["lightTransform"].tobytes()  # Space at frame 6 shows at 7
    assert rec[5]["ubo"]["view"].tobytes() != rec[3]["ubo"]["view"].tobytes()
    # the image: the oracle fed with the traced frames' blocks, reference image semantics (E restarts when iteration <= 1)
    scene = hostapi.load_model(obj)
    o = po.Oracle(scene, seed=1)
    acc = np.zeros((H, W, 4), np.float32)
    for r, e in zip(rec, exp):
        if r["traced"]:
            acc, _, _ = o.render(r["ubo"], W, H, e["mode"], accum=acc, flags=1)
    nan = np.isnan(gpu)
    assert (nan == np.isnan(acc)).all() and np.where(nan, 0, gpu).tobytes() == np.where(nan, 0, acc).tobytes()


def test_cli_aa_uses_the_frame_loops_stride(po, tmp_path):
    """prt_render --aa N: iteration k starts at Sobol index 1 + k*N (vkContext.cpp:1033), as N renderFrame() calls would."""
    import subprocess
    exe = os.path.join(os.path.dirname(hostapi.LIB_PATH), "prt_render")
    obj = os.path.join(G, "obj", "mixed.obj")
    pfm = tmp_path / "aa.pfm"
    W, H, n, aa = 40, 24, 3, 3
    subprocess.check_call([exe, obj, "--width", str(W), "--height", str(H), "--spp", str(n), "--aa", str(aa), "--bounces", "2",
                           "--eye", "0.2", "1.1", "0.9", "--target", "0", "0.6", "-0.4", "--out", "", "--pfm", str(pfm)])
    s = hostapi.default_settings(numIndicesBounces=2, samplesPerPixel=n, numAArays=aa)
    _, ubo = hostapi.render_obj(obj, W, H, s, (0.2, 1.1, 0.9), (0.0, 0.6, -0.4), first=1, n=1)
    o = po.Oracle(hostapi.load_model(obj), seed=1)
    acc = np.zeros((H, W, 4), np.float32)
    for k in range(n):
        u = ubo.copy()
        u["iteration"] = 1 + k * aa
        acc, _, _ = o.render(u, W, H, 0, accum=acc)
    raw = open(pfm, "rb").read()
    hdr_end = 0
    for _ in range(3):
        hdr_end = raw.index(b"\n", hdr_end) + 1
    img = np.frombuffer(raw[hdr_end:], "<f4").reshape(H, W, 3)[::-1]  # PFM rows are bottom-up; colour / w
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = acc[..., :3] / acc[..., 3:4]
    ok = np.isfinite(ref).all(axis=-1) & np.isfinite(img).all(axis=-1)
    assert ok.mean() > 0.5 and np.allclose(img[ok], ref[ok], rtol=0, atol=0)


def test_bench_renders_an_obj_file(po, tmp_path):
    """`bench.py --obj PATH`: a real asset goes through libprt_host's Model (OBJ + MTL + textures) into the HIP path; the
    line names the file, and the image of the timed region is the oracle's for the same scene and uniforms, bit for bit."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    from pathtracer_amd import capi, uniforms

    obj = os.path.join(root, "tests", "golden", "obj", "mixed.obj")
    W, H, K = 96, 64, 3
    out = tmp_path / "img.npy"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--obj", obj, "--width", str(W), "--height", str(H),
                          "--steps", str(K), "--warmup", "1", "--bounces", "4", "--no-cpu-baseline", "--dump-accum", str(out)],
                         cwd=root, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert "mixed.obj" in line["metric"] and "mixed.obj" in line["config"]["workload"] and "libprt_host" in line["data"]
    assert line["config"]["triangles"] == 16 and line["roofline"]["traffic"] is None  # no counters for this workload
    sc = bench.load_obj_scene(obj)
    assert len(sc["textures"]) == 5
    o = po.Oracle(sc, seed=2)
    acc, rays = np.zeros((H, W, 4), np.float32), 0
    for it in range(1, K + 1):
        u = uniforms.make_uniforms(sc, W, H, bounces=4, spp=1024, iteration=it, light_area=0.02)
        acc, _, r = o.render(u, W, H, capi.MODE_PATH, accum=acc)
        rays += r
    got = np.load(out)
    assert line["config"]["rays_total"] == rays
    assert got.tobytes() == acc.tobytes()
