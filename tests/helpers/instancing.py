"""Test helper: flatten a two-level scene (scenes.instanced_scene) into a single-level one the way
pathtracer_amd/host/Scene.cpp does — p = (M0*x + M1*y) + (M2*z + M3), n = normalize(transpose(inverse(M)) * n) in fp32
with the same operation order (numpy float32 arithmetic is IEEE and unfused); identity instances are copied bit for
bit. The inverse is the oracle's restatement of glm::inverse (test infrastructure)."""
import numpy as np

from oracle import pyoracle

_ID = np.eye(4, dtype=np.float32).T.reshape(16)


def flatten(sc):
    blas, inst = sc["blas"], sc["instances"]
    V = sc["vertices"]
    out = []
    for it in inst:
        b = blas[int(it["blas"])]
        v = V[3 * int(b["first_triangle"]):3 * int(b["first_triangle"] + b["n_triangles"])].copy()
        M = np.asarray(it["transform"], np.float32)
        if M.tobytes() != _ID.tobytes():
            W = pyoracle.inverse4(M)
            p, n = v["p"].astype(np.float32), v["n"].astype(np.float32)
            q = np.empty_like(p)
            m = np.empty_like(n)
            for k in range(3):
                q[:, k] = (M[0 + k] * p[:, 0] + M[4 + k] * p[:, 1]) + (M[8 + k] * p[:, 2] + M[12 + k])
                # row k of transpose(inverse(M)) = column k of the inverse
                m[:, k] = (W[4 * k + 0] * n[:, 0] + W[4 * k + 1] * n[:, 1]) + W[4 * k + 2] * n[:, 2]
            d = (m[:, 0] * m[:, 0] + m[:, 1] * m[:, 1]) + m[:, 2] * m[:, 2]
            ok = d > 0
            inv = np.float32(1.0) / np.sqrt(d[ok], dtype=np.float32)
            m[ok] = m[ok] * inv[:, None]
            v["p"], v["n"] = q, m
        out.append(v)
    verts = np.concatenate(out)
    flat = {k: sc[k] for k in ("materials", "textures", "camera", "light") if k in sc}
    flat["vertices"] = verts
    flat["indices"] = np.arange(len(verts), dtype=np.uint32)
    flat["name"] = sc.get("name", "") + "_flat"
    return flat


def prim_bases(sc):
    n = sc["blas"]["n_triangles"][sc["instances"]["blas"]].astype(np.uint64)
    return np.concatenate([[0], np.cumsum(n)]).astype(np.uint64)
