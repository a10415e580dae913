"""The C++ host layer (pathtracer_amd/host: Model, CameraControls, AreaLight, the uniform block) against
values produced by the libraries the reference itself uses — glm, tinyobjloader, stb_image — captured by
oracle/ref_host_dump.cpp into tests/golden/host_golden.json (bit-exact bar: every float is compared by bits)."""
import json
import os
import re

import numpy as np
import pytest

from pathtracer_amd import hostapi, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(G, "host_golden.json")), parse_int=float)  # "-0" must stay a negative zero


def f32(x):
    return np.asarray(x, np.float32)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "prt_host.h")).read()
    declared = set(re.findall(r"\b(prth_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(hostapi.EXPORTS)
    L = hostapi.lib()
    for n in declared:
        assert hasattr(L, n), n


def test_camera_and_uniform_block_bit_exact(golden):
    for pose in golden["camera"]:
        ubo, proj = hostapi.camera_uniforms(pose["fov"], pose["aspect"], pose["near"], pose["far"], pose["pos"],
                                            pose["yaw"], pose["pitch"])
        assert proj.tobytes() == f32(pose["projection"]).tobytes()
        for k in ("view", "proj", "modelIT", "projViewInverse", "lightTransform"):
            assert ubo[k].tobytes() == f32(pose[k]).tobytes(), (k, ubo[k], pose[k])
        assert ubo["model"].tobytes() == np.eye(4, dtype=np.float32).tobytes()
    # settings -> block (vkContext.cpp:1009-1026), defaults of vkContext.h:161-192
    s = hostapi.default_settings()
    assert (s.numIndicesBounces, s.samplesPerPixel, s.numAArays, s.numAOrays, s.rtRenderingMode, s.iteration) == (3, 128, 1, 16, 0, 1)
    s = hostapi.default_settings(numIndicesBounces=5, lightSourceArea=0.3, lightE=42.0, numAArays=4, iteration=9)
    ubo, _ = hostapi.camera_uniforms(65.0, 1.5, 0.01, 100.0, (1, 2, 3), 30.0, -10.0, s)
    assert ubo["numIndirectBounces"] == 5 and ubo["numAArays"] == 4 and ubo["iteration"] == 9
    assert ubo["lightSize"].tolist() == [np.float32(0.3)] * 2 and ubo["lightE"].tolist() == [42.0] * 3
    assert ubo["lightSourceArea"] == np.float32(0.3) and ubo["samplesPerPixel"] == 128


def _same_obj(mine, ref):
    assert mine["ok"] == ref["ok"]
    for k in ("vertices", "normals", "texcoords", "colors"):
        assert f32(mine[k]).tobytes() == f32(ref[k]).tobytes(), k
    assert [s["name"] for s in mine["shapes"]] == [s["name"] for s in ref["shapes"]]
    for a, b in zip(mine["shapes"], ref["shapes"]):
        assert a["indices"] == b["indices"] and a["material_ids"] == b["material_ids"], a["name"]
    assert len(mine["materials"]) == len(ref["materials"])
    for a, b in zip(mine["materials"], ref["materials"]):
        for k, v in b.items():
            if isinstance(v, str):
                assert a[k] == v, (b["name"], k)
            else:
                assert f32(a[k]).tobytes() == f32(v).tobytes(), (b["name"], k)
    if "warn" in ref:  # the messages too (they carry line numbers)
        assert mine["warn"] == ref["warn"] and mine["err"] == ref["err"]


def test_obj_reader_equals_the_reference_loader(golden, tmp_path):
    out = tmp_path / "mine.json"
    rc = hostapi.lib().prth_obj_dump_json(os.path.join(G, "obj", "mixed.obj").encode(), (os.path.join(G, "obj") + "/").encode(),
                                          str(out).encode())
    assert rc == 0
    _same_obj(json.load(open(out), parse_int=float), golden["obj"])


# tests/golden/obj_cases/*.json = tinyobj::LoadObj output (oracle/ref_host_dump --obj) for hand-written files:
#   ngons    concave / clockwise / star-shaped / collinear-start / non-planar / degenerate / bow-tie polygons in all
#            three projection planes, faces with 1 and 2 corners
#   indices  negative (relative) indices, i/t and i//n forms, out-of-range indices, vertex colours (full, partial),
#            number forms (exponents, ".5", "1e", "+3", "-0.0", long fractions), first mtllib name missing, a material
#            before any newmtl, `d` before / after `Tr`, unknown usemtl names (with a trailing blank), repeated usemtl
#   groups   flush order of o / g / usemtl, `g` with several / no names, bare keywords, lines and points keeping a
#            shape alive, a trailing usemtl
#   textures texture-statement options (with too few arguments, at the end of a line), names with blanks and
#            backslashes, map_d, later statements overriding earlier ones
#   crlf     \r\n and lone \r line ends, no newline at the end
@pytest.mark.parametrize("case", ["ngons", "indices", "groups", "textures", "crlf"])
def test_obj_reader_cases_equal_the_reference_loader(case, tmp_path):
    d = os.path.join(G, "obj_cases")
    # the goldens were made with the paths below relative to the repository root (they appear in warnings)
    out = tmp_path / "mine.json"
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        hostapi.lib().prth_obj_dump_json(f"tests/golden/obj_cases/{case}.obj".encode(), b"tests/golden/obj_cases/", str(out).encode())
    finally:
        os.chdir(cwd)
    ref = json.load(open(os.path.join(d, case + ".json")), parse_int=float)["obj"]
    mine = json.load(open(out), parse_int=float)
    assert len(ref["shapes"]) > 0
    _same_obj(mine, ref)


def test_obj_reader_rejects_what_the_reference_loader_rejects(tmp_path):
    # (verified with oracle/_ref/ref_host_dump; 16 000 random files: oracle/fuzz_obj_vs_ref.py, no difference)
    for text in ("v 0 0 0\nf 1 0 1\n", "v 0 0 0\nf 1/1/1/1 1 1\n", "v 0 0 0\nl 1 x\n", "v 0 0 0\np 0\n",
                 "v 0 0 0\nf 1 1 1 # c\n", "v 0 0 0\nf 1/ \n"):
        p = tmp_path / "bad.obj"
        p.write_text(text)
        with pytest.raises(hostapi.HostError):
            hostapi.load_model(str(p))
    # ... and accepts what it accepts: a corner ending in '/' takes the number from the next word (atoi skips blanks)
    p = tmp_path / "odd.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0.25 0.5\nvn 0 0 1\nf 1/ 1 2 3\nf 1//  1 2 3\n")
    out = tmp_path / "odd.json"
    assert hostapi.lib().prth_obj_dump_json(str(p).encode(), b"", str(out).encode()) == 0
    sh = json.load(open(out))["shapes"][0]["indices"]
    assert sh[:3] == [[0, -1, 0], [0, -1, -1], [1, -1, -1]] and sh[6:9] == [[0, 0, -1], [0, -1, -1], [1, -1, -1]]


def _expected_model(ref):
    """Model::LoadModelFromFile (Model.cpp:61-220) applied to the reference loader's output."""
    paths, mats = [], []
    for m in ref["materials"]:
        r = scenes.make_material()
        r["ambient"], r["diffuse"], r["specular"] = m["ambient"], m["diffuse"], m["specular"]
        r["emission"], r["transmittance"] = m["emission"], m["transmittance"]
        r["dissolve"], r["ior"], r["illum"], r["metallic"] = m["dissolve"], m["ior"], int(m["illum"]), 0.0
        rough = np.float32(m["roughness"])
        r["shininess"] = 0.0 if rough == 0 else np.float32(1.0 - float(rough))
        ids = []
        for key in ("diffuse_texname", "specular_texname", "bump_texname"):
            name, tid = m[key], -1
            if name in paths:
                tid = paths.index(name)
            elif name:
                paths.append(name.replace("\\\\", "/"))
                tid = len(paths) - 1
            ids.append(tid)
        r["diffuseTextureID"], r["specularTextureID"], r["normalTextureID"] = ids
        mats.append(r)
    V, N, T, Cc = (f32(ref[k]) for k in ("vertices", "normals", "texcoords", "colors"))
    verts = []
    for s in ref["shapes"]:
        for k, (vi, ni, ti) in enumerate([[int(x) for x in t] for t in s["indices"]]):
            v = np.zeros((), scenes.VERTEX_DTYPE)
            v["p"] = V[3 * vi:3 * vi + 3]
            if len(N) and ni >= 0:
                v["n"] = N[3 * ni:3 * ni + 3]
            if len(T) and ti >= 0:
                v["t"] = (T[2 * ti], -T[2 * ti + 1])
            v["c"] = Cc[3 * vi:3 * vi + 3]
            mid = int(s["material_ids"][k // 3])
            v["materialID"] = mid if 0 <= mid < len(mats) else 0
            verts.append(v)
    return np.array(verts, scenes.VERTEX_DTYPE), np.array(mats, scenes.MATERIAL_DTYPE), paths


def test_model_equals_reference_semantics(golden):
    m = hostapi.load_model(os.path.join(G, "obj", "mixed.obj"))
    v, mats, paths = _expected_model(golden["obj"])
    assert m["vertices"].tobytes() == v.tobytes()
    assert (m["indices"] == np.arange(len(v), dtype=np.uint32)).all()  # Model.cpp:203
    got, exp = m["materials"].copy(), mats.copy()
    got["pad"], exp["pad"] = 0, 0  # uninitialised in the reference
    assert got.tobytes() == exp.tobytes()
    # the dedup compares the raw MTL name with the stored (rewritten) path: "textures\\\\wood.png" is stored twice
    assert m["texture_paths"] == paths and paths.count("textures/wood.png") == 2
    for path, tex in zip(m["texture_paths"], m["textures"]):
        g = golden["textures"][os.path.basename(path)]
        assert tex.shape == (int(g["height"]), int(g["width"]), 4)
        assert tex.tobytes() == np.array(g["rgba"], np.uint8).tobytes(), path


def test_model_without_mtl_and_normals(tmp_path):
    p = tmp_path / "tri.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1 3 4\n")
    m = hostapi.load_model(str(p))
    assert len(m["vertices"]) == 6 and len(m["materials"]) == 1
    d = m["materials"][0]  # Material() defaults, Model.h:27-40
    assert d["diffuse"].tolist() == [0.0, 1.0, 1.0] and d["dissolve"] == 1.0 and d["diffuseTextureID"] == -1
    assert (m["vertices"]["n"][:3] == np.float32([0, 0, 1])).all()  # flat normals (Model.cpp:207-220)
    assert (m["vertices"]["n"][3:] == np.float32([1, 0, 0])).all()
    assert (m["vertices"]["materialID"] == 0).all() and (m["vertices"]["c"] == 1.0).all()
    assert len(m["textures"]) == 1 and m["textures"][0].tolist() == [[[255, 255, 255, 255]]]  # Model.cpp:313-343
    with pytest.raises(hostapi.HostError):
        hostapi.load_model(str(tmp_path / "missing.obj"))
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nf 0 1 2\n")  # index 0 is invalid: the loader fails
    with pytest.raises(hostapi.HostError):
        hostapi.load_model(str(bad))


def test_texture_decoders_equal_the_reference_decoder(golden):
    d = os.path.join(G, "obj", "textures")
    assert len(golden["textures"]) >= 25
    for name, g in golden["textures"].items():
        img = hostapi.load_image(os.path.join(d, name))
        assert g["ok"] and img is not None, name
        assert img.shape == (int(g["height"]), int(g["width"]), 4), name
        assert img.tobytes() == np.array(g["rgba"], np.uint8).tobytes(), name


def test_undecodable_texture_falls_back_to_white(tmp_path):
    (tmp_path / "t.jpg").write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 64)
    (tmp_path / "trunc.png").write_bytes(open(os.path.join(G, "obj", "textures", "wood.png"), "rb").read()[:200])
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 1 1\nmap_Kd t.jpg\nmap_Ks trunc.png\nmap_Bump -bm 0.5 nothere.png\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n")
    m = hostapi.load_model(str(tmp_path / "m.obj"))
    assert m["texture_paths"] == ["t.jpg", "trunc.png", "nothere.png"]  # texture options are skipped
    assert all(t.tolist() == [[[255, 255, 255, 255]]] for t in m["textures"])  # Model.cpp:352-359
    assert m["materials"][0]["normalTextureID"] == 2


# ---- several models / instances (Scene.h; vkRT_TLAS.cpp:58-72) ---------------------------------------------------
def _glm_mul_point(M, p):
    """(M * vec4(p, 1)).xyz in glm's order: (m0*x + m1*y) + (m2*z + m3*1), one fp32 rounding per operation."""
    M = np.asarray(M, np.float32).reshape(4, 4)  # M[col][row]
    x, y, z = (p[:, k:k + 1].astype(np.float32) for k in range(3))
    a = (M[0][None, :] * x + M[1][None, :] * y).astype(np.float32)
    b = (M[2][None, :] * z + M[3][None, :] * np.float32(1)).astype(np.float32)
    return (a + b).astype(np.float32)[:, :3]


def test_flatten_instances(tmp_path):
    (tmp_path / "b.mtl").write_text("newmtl red\nKd 1 0 0\nmap_Kd nothere.png\nnewmtl blue\nKd 0 0 1\n")
    (tmp_path / "b.obj").write_text("mtllib b.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 -0\nvn 0 0 1\nusemtl red\nf 1//1 2//1 3//1\n"
                                    "usemtl blue\nf 4//1 3//1 2//1\n")
    a = hostapi.open_model(os.path.join(G, "obj", "mixed.obj"))
    b = hostapi.open_model(str(tmp_path / "b.obj"))
    sa, sb = a.scene(), b.scene()
    ident = np.eye(4, dtype=np.float32)
    # column-major: rotation about z by 90 deg, scale (2, 3, 0.5), translation (1, -2, 3)
    M = np.zeros((4, 4), np.float32)
    M[0] = (0, 2, 0, 0)
    M[1] = (-3, 0, 0, 0)
    M[2] = (0, 0, 0.5, 0)
    M[3] = (1, -2, 3, 1)
    flat, first = hostapi.flatten_instances([a, b], [(0, ident, 7), (1, M, 8), (1, ident, 9), (0, M, 10)])
    fs = flat.scene()
    na, nb = len(sa["vertices"]), len(sb["vertices"])
    assert first.tolist() == [0, na // 3, (na + nb) // 3, (na + 2 * nb) // 3, (2 * na + 2 * nb) // 3]
    assert (fs["indices"] == np.arange(2 * na + 2 * nb, dtype=np.uint32)).all()
    nma, nta = len(sa["materials"]), len(sa["textures"])
    assert len(fs["materials"]) == nma + len(sb["materials"]) and len(fs["textures"]) == nta + len(sb["textures"])
    # merged tables: model 0 verbatim, model 1 with texture ids rebased (-1 stays -1)
    ma, mb = fs["materials"][:nma].copy(), fs["materials"][nma:].copy()
    ea, eb = sa["materials"].copy(), sb["materials"].copy()
    for m in (ma, mb, ea, eb):
        m["pad"] = 0
    assert ma.tobytes() == ea.tobytes()
    for f in ("diffuseTextureID", "specularTextureID", "normalTextureID"):
        assert (mb[f] == np.where(eb[f] >= 0, eb[f] + nta, eb[f])).all()
        mb[f] = eb[f]
    assert mb.tobytes() == eb.tobytes()
    assert eb["diffuseTextureID"][0] == 0 and eb["diffuseTextureID"][1] == -1
    for t, e in zip(fs["textures"], sa["textures"] + sb["textures"]):
        assert t.tobytes() == e.tobytes()
    v = fs["vertices"]
    # identity instances are bit-for-bit copies (a negative zero survives), material ids rebased
    i0 = v[:na].copy()
    assert i0.tobytes() == sa["vertices"].tobytes()
    i2 = v[na + nb:na + 2 * nb].copy()
    assert (i2["materialID"] == sb["vertices"]["materialID"] + nma).all()
    i2["materialID"] -= nma
    assert i2.tobytes() == sb["vertices"].tobytes()
    assert np.signbit(sb["vertices"]["p"][3][2])
    # transformed instances: positions bit-exact in glm's order, normals = normalize(inverse-transpose * n)
    for sl, src, base in ((slice(na, na + nb), sb, nma), (slice(na + 2 * nb, 2 * na + 2 * nb), sa, 0)):
        got = v[sl]
        assert got["p"].tobytes() == _glm_mul_point(M, src["vertices"]["p"]).tobytes()
        assert (got["materialID"] == src["vertices"]["materialID"] + base).all()
        assert got["t"].tobytes() == src["vertices"]["t"].tobytes() and got["c"].tobytes() == src["vertices"]["c"].tobytes()
        A = M.T[:3, :3].astype(np.float64)  # row-major linear part
        n = src["vertices"]["n"].astype(np.float64) @ np.linalg.inv(A)  # == (inverse-transpose(A) * n) per row
        ln = np.linalg.norm(n, axis=1, keepdims=True)
        exp = np.where(ln > 0, n / np.where(ln > 0, ln, 1), 0)
        ok = np.isfinite(exp).all(axis=1)
        assert np.allclose(got["n"][ok], exp[ok], atol=2e-6)
    with pytest.raises(hostapi.HostError):
        hostapi.flatten_instances([a], [(3, ident, 0)])
    with pytest.raises(hostapi.HostError):
        hostapi.flatten_instances([a], [])
    for m in (a, b, flat):
        m.close()


def test_png_writer_round_trip(tmp_path):
    import struct
    import zlib
    rng = np.random.default_rng(4)
    for (h, w) in ((1, 1), (3, 5), (130, 131), (257, 300)):  # the last two span several stored blocks
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        p = str(tmp_path / f"o{h}.png")
        hostapi.write_png(p, img)
        back = hostapi.load_image(p)  # the stb-equivalent decoder of the host layer
        assert back.tobytes() == img.tobytes()
        raw = open(p, "rb").read()  # and an independent check of the container: chunk CRCs, zlib stream
        assert raw[:8] == b"\x89PNG\r\n\x1a\n"
        pos, idat = 8, b""
        while pos < len(raw):
            n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
            body = raw[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
            if typ == b"IHDR":
                assert struct.unpack(">IIBBBBB", body) == (w, h, 8, 6, 0, 0, 0)
            if typ == b"IDAT":
                idat += body
            pos += 12 + n
        rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * 4 + 1)
        assert (rows[:, 0] == 0).all() and rows[:, 1:].tobytes() == img.tobytes()


def test_checkpoint_file_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    acc = rng.standard_normal((7, 9, 4)).astype(np.float32)
    acc[0, 0, 0] = np.nan  # the reference's NaN pixels must survive a save / restore unchanged
    p = str(tmp_path / "s.ckpt")
    hostapi.checkpoint_write(p, acc, 17, mode=0, bounces=5, aa_rays=1)
    c = hostapi.checkpoint_read(p)
    assert c["accum"].tobytes() == acc.tobytes()
    assert (c["next_iteration"], c["mode"], c["bounces"], c["aa_rays"]) == (17, 0, 5, 1)
    raw = bytearray(open(p, "rb").read())
    assert len(raw) == 64 + acc.nbytes and raw[:8] == b"PRTCKPT1"
    bad = str(tmp_path / "bad.ckpt")
    for mutate in (lambda r: r[:-4], lambda r: r + b"\0", lambda r: b"XRTCKPT1" + r[8:],
                   lambda r: r[:100] + bytes([r[100] ^ 1]) + r[101:]):
        open(bad, "wb").write(mutate(bytes(raw)))
        with pytest.raises(hostapi.HostError):
            hostapi.checkpoint_read(bad)
    with pytest.raises(hostapi.HostError):
        hostapi.checkpoint_read(str(tmp_path / "missing.ckpt"))


def test_malformed_textures_fall_back_without_huge_allocations(tmp_path):
    """a PNG header that claims 2^24 x 2^24 pixels, a stream that inflates far beyond its image, a colour-mapped
    TGA without palette, truncated files: each is refused (Model then uses the reference's 1x1 white texture, Model.cpp:352-359)
    and none of them makes the decoder allocate what the header claims."""
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    def png(w, h, idat, ctype=6, depth=8):
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + chunk(b"IDAT", idat) + chunk(b"IEND", b"")

    good_raw = b"".join(b"\0" + bytes([10 * y, 20, 30, 255] * 3) for y in range(2))
    cases = {
        "huge_header.png": png(1 << 24, 1 << 24, zlib.compress(good_raw)),
        "big_header_small_data.png": png(16000, 16000, zlib.compress(good_raw)),
        "truncated_stream.png": png(3, 2, zlib.compress(good_raw)[:-9]),
        "nopalette.tga": bytes([0, 1, 1, 0, 0, 0, 0, 24, 0, 0, 0, 0, 2, 0, 2, 0, 8, 0]) + b"\0\1\2\3",
        "short.tga": bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 0, 4, 0, 24, 0]) + b"\0" * 5,
    }
    for name, data in cases.items():
        (tmp_path / name).write_bytes(data)
        assert hostapi.load_image(str(tmp_path / name)) is None, name
    # a bomb: 3 x 2 pixels, but the stream inflates to 64 MiB -> decoded from its first 26 bytes, the rest never produced
    (tmp_path / "bomb.png").write_bytes(png(3, 2, zlib.compress(good_raw + b"\0" * (64 << 20))))
    img = hostapi.load_image(str(tmp_path / "bomb.png"))
    assert img is not None and img.shape == (2, 3, 4) and img[1, 0].tolist() == [10, 20, 30, 255]
    (tmp_path / "ok.png").write_bytes(png(3, 2, zlib.compress(good_raw)))
    assert hostapi.load_image(str(tmp_path / "ok.png")).tobytes() == img.tobytes()
