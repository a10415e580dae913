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


def test_obj_reader_equals_the_reference_loader(golden, tmp_path):
    out = tmp_path / "mine.json"
    rc = hostapi.lib().prth_obj_dump_json(os.path.join(G, "obj", "mixed.obj").encode(), (os.path.join(G, "obj") + "/").encode(),
                                          str(out).encode())
    assert rc == 0
    mine, ref = json.load(open(out), parse_int=float), golden["obj"]
    for k in ("vertices", "normals", "texcoords", "colors"):
        assert f32(mine[k]).tobytes() == f32(ref[k]).tobytes(), k
    assert [s["name"] for s in mine["shapes"]] == [s["name"] for s in ref["shapes"]]
    for a, b in zip(mine["shapes"], ref["shapes"]):
        assert a["indices"] == b["indices"] and a["material_ids"] == b["material_ids"]
    assert len(mine["materials"]) == len(ref["materials"])
    for a, b in zip(mine["materials"], ref["materials"]):
        for k, v in b.items():
            if isinstance(v, str):
                assert a[k] == v, (b["name"], k)
            else:
                assert f32(a[k]).tobytes() == f32(v).tobytes(), (b["name"], k)


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
