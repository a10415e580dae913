"""Procedural stand-ins for the reference's scenes.

The reference hard-codes OBJ paths under an untracked ``scenes/`` directory
(/root/reference/src/vkContext.cpp:52-72, .gitignore:21) and there is no network, so the named
assets (Cornell box, conference room, Crytek Sponza, San Miguel) cannot be loaded here. These
generators emit geometry of the same scale and character, already in the layout that
``Model::LoadModelFromFile`` (/root/reference/src/Model.cpp:47-221) produces: one de-indexed
``VertexPNTC`` per triangle corner, identity ``uint32`` indices, ``Material`` records with the
loader's ``shininess = 1 - Pr`` convention, RGBA8 textures.

Everything is deterministic in the integer ``seed``.
"""
from __future__ import annotations

import numpy as np

VERTEX_DTYPE = np.dtype(
    [("p", "<f4", 3), ("n", "<f4", 3), ("t", "<f4", 2), ("c", "<f4", 3), ("materialID", "<i4")]
)
MATERIAL_DTYPE = np.dtype(
    [
        ("ambient", "<f4", 3),
        ("diffuse", "<f4", 3),
        ("specular", "<f4", 3),
        ("transmittance", "<f4", 3),
        ("emission", "<f4", 3),
        ("shininess", "<f4"),
        ("metallic", "<f4"),
        ("ior", "<f4"),
        ("dissolve", "<f4"),
        ("illum", "<i4"),
        ("diffuseTextureID", "<i4"),
        ("specularTextureID", "<i4"),
        ("normalTextureID", "<i4"),
        ("pad", "<f4"),
    ]
)
# include/prt_types.h: prt_blas_range / prt_instance (transform: column-major mat4, object -> world)
BLAS_DTYPE = np.dtype([("first_triangle", "<u4"), ("n_triangles", "<u4")])
INSTANCE_DTYPE = np.dtype([("transform", "<f4", 16), ("blas", "<u4"), ("instance_id", "<u4"), ("hit_group", "<u4"),
                           ("pad", "<u4")])
assert VERTEX_DTYPE.itemsize == 48 and MATERIAL_DTYPE.itemsize == 96 and INSTANCE_DTYPE.itemsize == 80


def make_material(diffuse=(0.8, 0.8, 0.8), specular=(0.0, 0.0, 0.0), roughness=None, dissolve=1.0,
                  emission=(0.0, 0.0, 0.0), tex=(-1, -1, -1)):
    """One Material record with the reference loader's conventions (Model.cpp:74-150):
    ``shininess = 0`` when the MTL has no ``Pr``, else ``1 - Pr``."""
    m = np.zeros((), MATERIAL_DTYPE)
    m["ambient"] = (0.1, 0.1, 0.1)
    m["diffuse"] = diffuse
    m["specular"] = specular
    m["transmittance"] = (0.0, 0.0, 0.0)
    m["emission"] = emission
    m["shininess"] = 0.0 if roughness is None or roughness == 0.0 else 1.0 - roughness
    m["metallic"] = 0.0
    m["ior"] = 1.0
    m["dissolve"] = dissolve
    m["illum"] = 2
    m["diffuseTextureID"], m["specularTextureID"], m["normalTextureID"] = tex
    return m


class MeshBuilder:
    """Accumulates triangle soups (P: [n,3,3] float32 positions, optional normals / uvs)."""

    def __init__(self):
        self.P, self.N, self.T, self.M = [], [], [], []

    def add(self, P, mat, N=None, T=None):
        P = np.asarray(P, np.float32).reshape(-1, 3, 3)
        if N is None:  # flat normals, as the loader generates when the file has none (Model.cpp:207-220)
            e1 = P[:, 1] - P[:, 0]
            e2 = P[:, 2] - P[:, 0]
            n = np.cross(e1, e2)
            ln = np.linalg.norm(n, axis=1, keepdims=True)
            n = n / np.where(ln > 0, ln, 1.0)
            N = np.repeat(n[:, None, :], 3, axis=1)
        if T is None:
            T = np.zeros((P.shape[0], 3, 2), np.float32)
        self.P.append(P)
        self.N.append(np.asarray(N, np.float32).reshape(-1, 3, 3))
        self.T.append(np.asarray(T, np.float32).reshape(-1, 3, 2))
        self.M.append(np.full(P.shape[0], mat, np.int32))

    def quad_grid(self, origin, eu, ev, nu, nv, mat, uv_scale=1.0, flip=False, displace=None):
        """A parallelogram origin + s*eu + t*ev tessellated into nu x nv cells (2 triangles each)."""
        origin, eu, ev = (np.asarray(a, np.float64) for a in (origin, eu, ev))
        s = np.linspace(0.0, 1.0, nu + 1)
        t = np.linspace(0.0, 1.0, nv + 1)
        S, Tt = np.meshgrid(s, t, indexing="ij")
        pts = origin + S[..., None] * eu + Tt[..., None] * ev
        if displace is not None:
            pts = pts + displace(S, Tt)
        uv = np.stack([S * uv_scale, Tt * uv_scale], -1)
        a, b, c, d = pts[:-1, :-1], pts[1:, :-1], pts[1:, 1:], pts[:-1, 1:]
        ua, ub, uc, ud = uv[:-1, :-1], uv[1:, :-1], uv[1:, 1:], uv[:-1, 1:]
        if flip:
            tri = np.stack([np.stack([a, c, b], -2), np.stack([a, d, c], -2)], 2)
            tuv = np.stack([np.stack([ua, uc, ub], -2), np.stack([ua, ud, uc], -2)], 2)
        else:
            tri = np.stack([np.stack([a, b, c], -2), np.stack([a, c, d], -2)], 2)
            tuv = np.stack([np.stack([ua, ub, uc], -2), np.stack([ua, uc, ud], -2)], 2)
        self.add(tri.reshape(-1, 3, 3), mat, T=tuv.reshape(-1, 3, 2))

    def box(self, lo, hi, mat, n=1, rot_y=0.0):
        lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
        c = 0.5 * (lo + hi)
        sub = MeshBuilder()
        dx, dy, dz = hi - lo
        X, Y, Z = np.array([dx, 0, 0.0]), np.array([0, dy, 0.0]), np.array([0, 0, dz])
        sub.quad_grid(lo, Z, Y, n, n, mat)  # -x
        sub.quad_grid(lo + X, Y, Z, n, n, mat)  # +x
        sub.quad_grid(lo, X, Z, n, n, mat)  # -y
        sub.quad_grid(lo + Y, Z, X, n, n, mat)  # +y
        sub.quad_grid(lo, Y, X, n, n, mat)  # -z
        sub.quad_grid(lo + Z, X, Y, n, n, mat)  # +z
        P = np.concatenate(sub.P)
        T = np.concatenate(sub.T)
        if rot_y != 0.0:
            cs, sn = np.cos(rot_y), np.sin(rot_y)
            R = np.array([[cs, 0, sn], [0, 1, 0], [-sn, 0, cs]])
            P = ((P.reshape(-1, 3) - c) @ R.T + c).reshape(-1, 3, 3)
        self.add(P, mat, T=T)

    def sphere(self, center, radius, nu, nv, mat, smooth=True, squash=(1.0, 1.0, 1.0)):
        center = np.asarray(center, np.float64)
        th = np.linspace(0.0, np.pi, nv + 1)
        ph = np.linspace(0.0, 2 * np.pi, nu + 1)
        TH, PH = np.meshgrid(th, ph, indexing="ij")
        n = np.stack([np.sin(TH) * np.cos(PH), np.cos(TH), np.sin(TH) * np.sin(PH)], -1)
        pts = center + radius * n * np.asarray(squash)
        uv = np.stack([PH / (2 * np.pi), TH / np.pi], -1)
        a, b, c, d = pts[:-1, :-1], pts[1:, :-1], pts[1:, 1:], pts[:-1, 1:]
        na, nb, nc, nd = n[:-1, :-1], n[1:, :-1], n[1:, 1:], n[:-1, 1:]
        ua, ub, uc, ud = uv[:-1, :-1], uv[1:, :-1], uv[1:, 1:], uv[:-1, 1:]
        tri = np.stack([np.stack([a, c, b], -2), np.stack([a, d, c], -2)], 2).reshape(-1, 3, 3)
        nrm = np.stack([np.stack([na, nc, nb], -2), np.stack([na, nd, nc], -2)], 2).reshape(-1, 3, 3)
        tuv = np.stack([np.stack([ua, uc, ub], -2), np.stack([ua, ud, uc], -2)], 2).reshape(-1, 3, 2)
        # drop the degenerate triangles at the poles
        e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
        keep = np.linalg.norm(np.cross(e1, e2), axis=1) > 1e-12
        self.add(tri[keep], mat, N=nrm[keep] if smooth else None, T=tuv[keep])

    def cylinder(self, base, radius, height, n, mat, rings=1):
        base = np.asarray(base, np.float64)
        ph = np.linspace(0.0, 2 * np.pi, n + 1)
        ys = np.linspace(0.0, height, rings + 1)
        PH, Y = np.meshgrid(ph, ys, indexing="ij")
        nrm = np.stack([np.cos(PH), np.zeros_like(PH), np.sin(PH)], -1)
        pts = base + radius * nrm + np.stack([np.zeros_like(Y), Y, np.zeros_like(Y)], -1)
        a, b, c, d = pts[:-1, :-1], pts[1:, :-1], pts[1:, 1:], pts[:-1, 1:]
        na, nb, nc, nd = nrm[:-1, :-1], nrm[1:, :-1], nrm[1:, 1:], nrm[:-1, 1:]
        tri = np.stack([np.stack([a, c, b], -2), np.stack([a, d, c], -2)], 2).reshape(-1, 3, 3)
        nn = np.stack([np.stack([na, nc, nb], -2), np.stack([na, nd, nc], -2)], 2).reshape(-1, 3, 3)
        self.add(tri, mat, N=nn)
        # caps
        top = base + np.array([0, height, 0.0])
        ring = radius * np.stack([np.cos(ph), np.zeros_like(ph), np.sin(ph)], -1)
        capT = np.stack([np.broadcast_to(top, (n, 3)), top + ring[1:], top + ring[:-1]], 1)
        capB = np.stack([np.broadcast_to(base, (n, 3)), base + ring[:-1], base + ring[1:]], 1)
        self.add(np.concatenate([capT, capB]), mat)

    def finish(self, materials, textures=()):
        P = np.concatenate(self.P)
        N = np.concatenate(self.N)
        T = np.concatenate(self.T)
        M = np.concatenate(self.M)
        nt = P.shape[0]
        v = np.zeros(nt * 3, VERTEX_DTYPE)
        v["p"] = P.reshape(-1, 3)
        v["n"] = N.reshape(-1, 3)
        v["t"] = T.reshape(-1, 2)
        v["c"] = 1.0
        v["materialID"] = np.repeat(M, 3)
        idx = np.arange(nt * 3, dtype=np.uint32)  # Model.cpp:203: indices[i] = i
        mats = np.array(materials, MATERIAL_DTYPE)
        return {"vertices": v, "indices": idx, "materials": mats, "textures": list(textures)}


# ----------------------------------------------------------------------------------------------
def _lookat_pose(eye, target):
    return {"eye": np.asarray(eye, np.float32), "target": np.asarray(target, np.float32)}


def cornell_box():
    """BASELINE.json configs[0]: Cornell box, 36 triangles (5 walls, 2 boxes, ceiling light quad)."""
    mats = [
        make_material((0.73, 0.73, 0.73)),  # 0 white
        make_material((0.65, 0.05, 0.05)),  # 1 red
        make_material((0.12, 0.45, 0.15)),  # 2 green
        make_material((0.78, 0.78, 0.78), emission=(15.0, 15.0, 15.0)),  # 3 light
    ]
    mb = MeshBuilder()
    # room [-1,1] x [0,2] x [-1,1], open towards +z
    mb.quad_grid((-1, 0, -1), (0, 0, 2), (2, 0, 0), 1, 1, 0)  # floor (normal +y)
    mb.quad_grid((-1, 2, -1), (2, 0, 0), (0, 0, 2), 1, 1, 0)  # ceiling (normal -y)
    mb.quad_grid((-1, 0, -1), (2, 0, 0), (0, 2, 0), 1, 1, 0)  # back wall (normal +z)
    mb.quad_grid((-1, 0, -1), (0, 2, 0), (0, 0, 2), 1, 1, 1)  # left wall, red (normal +x)
    mb.quad_grid((1, 0, -1), (0, 0, 2), (0, 2, 0), 1, 1, 2)  # right wall, green (normal -x)
    mb.box((-0.65, 0.0, -0.1), (-0.05, 0.6, 0.5), 0, rot_y=0.3)  # short box
    mb.box((0.1, 0.0, -0.7), (0.7, 1.2, -0.1), 0, rot_y=-0.3)  # tall box
    mb.quad_grid((-0.25, 1.995, -0.25), (0.5, 0, 0), (0, 0, 0.5), 1, 1, 3)  # light quad
    sc = mb.finish(mats)
    sc["name"] = "cornell"
    sc["camera"] = _lookat_pose((0.0, 1.0, 2.5), (0.0, 1.0, 0.0))
    sc["light"] = {"pos": (0.0, 1.95, 0.0), "dir": (0.0, -1.0, 0.0)}
    return sc


def _value_noise_texture(rng, size, base, amp, octaves=4):
    img = np.zeros((size, size), np.float64)
    for o in range(octaves):
        n = 4 << o
        g = rng.random((n, n))
        reps = size // n
        up = np.kron(g, np.ones((reps, reps)))
        # cheap smoothing so bilinear taps see gradients
        up = 0.25 * (up + np.roll(up, 1, 0) + np.roll(up, 1, 1) + np.roll(np.roll(up, 1, 0), 1, 1))
        img += up / (1 << o)
    img = img / img.max()
    rgb = np.clip(np.asarray(base)[None, None, :] * (1.0 - amp + amp * img[..., None]), 0, 1)
    out = np.empty((size, size, 4), np.uint8)
    out[..., :3] = (rgb * 255.0 + 0.5).astype(np.uint8)
    out[..., 3] = 255
    return np.ascontiguousarray(out)


def _normal_map_texture(rng, size):
    h = _value_noise_texture(rng, size, (1, 1, 1), 1.0)[..., 0].astype(np.float64) / 255.0
    dx = np.roll(h, -1, 1) - np.roll(h, 1, 1)
    dy = np.roll(h, -1, 0) - np.roll(h, 1, 0)
    n = np.stack([-dx * 2.0, -dy * 2.0, np.ones_like(h)], -1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    out = np.empty((size, size, 4), np.uint8)
    out[..., :3] = ((n * 0.5 + 0.5) * 255.0 + 0.5).astype(np.uint8)
    out[..., 3] = 255
    return np.ascontiguousarray(out)


def conference_room(seed=2, detail=1.4, textured=False):
    """BASELINE.json configs[1]: conference-room-scale closed interior, ~330k triangles at the default detail=1.4.

    A 12 x 3.2 x 8 m closed room with tessellated walls, a long table, two rows of chairs with curved
    backs, ceiling fixtures, shelves with books, glass and glossy decoration. ``textured=True`` adds
    diffuse / specular / normal textures (Sponza-like material divergence, configs[3])."""
    rng = np.random.default_rng(seed)
    q = lambda n: max(1, int(round(n * detail)))
    textures = []
    tex = lambda d=-1, s=-1, n=-1: (d, s, n)
    if textured:
        textures = [
            _value_noise_texture(rng, 256, (0.75, 0.72, 0.68), 0.5),  # 0 plaster
            _value_noise_texture(rng, 256, (0.55, 0.35, 0.2), 0.6),  # 1 wood
            _value_noise_texture(rng, 128, (0.4, 0.42, 0.5), 0.7),  # 2 carpet
            _value_noise_texture(rng, 128, (0.9, 0.9, 0.9), 0.8),  # 3 spec map
            _normal_map_texture(rng, 256),  # 4 normal map
            _normal_map_texture(rng, 128),  # 5 normal map
        ]
    mats = [
        make_material((0.75, 0.73, 0.7), tex=tex(0 if textured else -1, -1, 4 if textured else -1)),  # 0 walls
        make_material((0.35, 0.36, 0.42), tex=tex(2 if textured else -1, -1, 5 if textured else -1)),  # 1 carpet
        make_material((0.45, 0.28, 0.15), (0.3, 0.3, 0.3), roughness=0.25,
                      tex=tex(1 if textured else -1, 3 if textured else -1, -1)),  # 2 table wood, glossy
        make_material((0.12, 0.12, 0.14), (0.6, 0.6, 0.6), roughness=0.15),  # 3 chair frame metal
        make_material((0.55, 0.1, 0.1), (0.05, 0.05, 0.05), roughness=0.7),  # 4 chair fabric
        make_material((0.9, 0.9, 0.9), (0.04, 0.04, 0.04), roughness=0.05, dissolve=0.0),  # 5 glass
        make_material((0.8, 0.8, 0.75), emission=(2.0, 2.0, 1.8)),  # 6 fixture (emissive)
        make_material((0.2, 0.3, 0.6), (0.2, 0.2, 0.2), roughness=0.4, dissolve=0.5),  # 7 "dissolve" branch
        make_material((0.7, 0.6, 0.2), (0.8, 0.7, 0.3), roughness=0.1),  # 8 brass
        make_material((0.3, 0.5, 0.3), roughness=0.9),  # 9 books / plants
    ]
    mb = MeshBuilder()
    W, H, D = 12.0, 3.2, 8.0
    x0, x1, z0, z1 = -W / 2, W / 2, -D / 2, D / 2
    g = q(48)
    mb.quad_grid((x0, 0, z0), (0, 0, D), (W, 0, 0), g, g, 1, uv_scale=6)  # floor
    mb.quad_grid((x0, H, z0), (W, 0, 0), (0, 0, D), g, g, 0, uv_scale=4)  # ceiling
    mb.quad_grid((x0, 0, z0), (W, 0, 0), (0, H, 0), g, q(16), 0, uv_scale=4)  # back  (z0)
    mb.quad_grid((x0, 0, z1), (0, H, 0), (W, 0, 0), q(16), g, 0, uv_scale=4)  # front (z1)
    mb.quad_grid((x0, 0, z0), (0, H, 0), (0, 0, D), q(16), g, 0, uv_scale=4)  # left
    mb.quad_grid((x1, 0, z0), (0, 0, D), (0, H, 0), g, q(16), 0, uv_scale=4)  # right
    # table: slab + legs
    mb.box((-3.6, 0.72, -1.0), (3.6, 0.78, 1.0), 2, n=q(24))
    for sx in (-3.3, 3.3):
        for sz in (-0.8, 0.8):
            mb.cylinder((sx, 0.0, sz), 0.05, 0.72, q(24), 3, rings=q(4))
    # chairs: two rows
    nchairs = 12
    for row, (zc, face) in enumerate(((-1.7, 1.0), (1.7, -1.0))):
        for i in range(nchairs):
            xc = -3.3 + 6.6 * i / (nchairs - 1) + rng.normal(0, 0.02)
            zz = zc + rng.normal(0, 0.03)
            mb.box((xc - 0.22, 0.42, zz - 0.22), (xc + 0.22, 0.47, zz + 0.22), 4, n=q(6))
            for lx in (-0.19, 0.19):
                for lz in (-0.19, 0.19):
                    mb.cylinder((xc + lx, 0.0, zz + lz), 0.015, 0.42, q(12), 3, rings=q(2))
            # curved back rest
            zb = zz - face * 0.22

            def bend(S, T, face=face):
                return np.stack([np.zeros_like(S), np.zeros_like(S), -face * 0.06 * np.sin(np.pi * S)], -1)

            mb.quad_grid((xc - 0.22, 0.5, zb), (0.44, 0, 0), (0, 0.45, 0), q(20), q(20), 4,
                         flip=(face < 0), displace=bend)
            mb.sphere((xc, 0.5, zz), 0.2, q(24), q(12), 4, squash=(1.0, 0.25, 1.0))  # cushion
    # ceiling fixtures
    for fx in np.linspace(-4.0, 4.0, 5):
        for fz in (-2.0, 2.0):
            mb.cylinder((fx, H - 0.12, fz), 0.35, 0.1, q(48), 6, rings=1)
    # decoration on the table: glass spheres, brass spheres, "dissolve" spheres
    for i in range(10):
        xc = -3.0 + 6.0 * i / 9
        m = (5, 8, 7)[i % 3]
        r = 0.09 + 0.03 * rng.random()
        mb.sphere((xc, 0.78 + r, rng.normal(0, 0.25)), r, q(64), q(32), m)
    # shelves with books along the back wall
    for sx in np.linspace(-5.0, 5.0, 6):
        mb.box((sx - 0.7, 0.0, z0 + 0.02), (sx + 0.7, 2.2, z0 + 0.4), 2, n=q(4))
        for level in range(4):
            y = 0.25 + 0.5 * level
            xb = sx - 0.62
            while xb < sx + 0.55:
                wbk = 0.03 + 0.05 * rng.random()
                hbk = 0.25 + 0.15 * rng.random()
                mb.box((xb, y, z0 + 0.42), (xb + wbk, y + hbk, z0 + 0.62), 9, n=q(2),
                       rot_y=rng.normal(0, 0.05))
                xb += wbk + 0.004
    # plants in the corners: many small spheres
    for cx, cz in ((x0 + 0.6, z1 - 0.6), (x1 - 0.6, z1 - 0.6)):
        mb.cylinder((cx, 0.0, cz), 0.22, 0.45, q(32), 8, rings=q(3))
        for _ in range(q(40)):
            off = rng.normal(0, 0.22, 3)
            mb.sphere((cx + off[0], 1.0 + abs(off[1]) * 1.5, cz + off[2]), 0.07, q(16), q(8), 9)
    sc = mb.finish(mats, textures)
    sc["name"] = "conference" + ("_textured" if textured else "")
    sc["camera"] = _lookat_pose((-5.2, 1.7, 3.3), (1.5, 0.8, -1.0))
    sc["light"] = {"pos": (0.0, H - 0.3, 0.0), "dir": (0.0, -1.0, 0.0)}
    return sc


def random_soup(n_tris, seed=7, extent=4.0, size=0.6):
    """Unstructured triangle soup for unit tests (BVH invariants, traversal vs brute force)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n_tris, 1, 3))
    P = c + rng.normal(0, size, (n_tris, 3, 3))
    mb = MeshBuilder()
    mb.add(P, 0)
    sc = mb.finish([make_material()])
    sc["name"] = f"soup{n_tris}"
    sc["camera"] = _lookat_pose((0.0, 0.0, extent * 2.5), (0.0, 0.0, 0.0))
    sc["light"] = {"pos": (0.0, extent * 1.5, 0.0), "dir": (0.0, -1.0, 0.0)}
    return sc


def sponza_scale(seed=4, detail=1.0):
    """BASELINE.json configs[3] stand-in: an atrium of Crytek-Sponza scale (~260k triangles at detail=1), every
    surface textured (diffuse + specular + normal maps), two storeys of columns and arches, hanging drapes —
    long thin occluders and many material switches per path, i.e. divergence-heavy secondary rays."""
    rng = np.random.default_rng(seed)
    q = lambda n: max(1, int(round(n * detail)))
    tex = [
        _value_noise_texture(rng, 256, (0.70, 0.66, 0.58), 0.6),  # 0 stone
        _value_noise_texture(rng, 256, (0.55, 0.50, 0.45), 0.7),  # 1 floor
        _value_noise_texture(rng, 128, (0.65, 0.15, 0.12), 0.5),  # 2 drape red
        _value_noise_texture(rng, 128, (0.15, 0.25, 0.60), 0.5),  # 3 drape blue
        _value_noise_texture(rng, 128, (0.85, 0.85, 0.85), 0.8),  # 4 specular map
        _normal_map_texture(rng, 256),  # 5
        _normal_map_texture(rng, 128),  # 6
        _value_noise_texture(rng, 64, (0.30, 0.45, 0.20), 0.8),  # 7 plants
    ]
    mats = [
        make_material((0.9, 0.9, 0.9), (0.05, 0.05, 0.05), roughness=0.8, tex=(0, -1, 5)),  # 0 stone walls
        make_material((0.9, 0.9, 0.9), (0.25, 0.25, 0.25), roughness=0.3, tex=(1, 4, 6)),  # 1 polished floor
        make_material((0.9, 0.9, 0.9), (0.02, 0.02, 0.02), roughness=0.9, tex=(2, -1, -1)),  # 2 drape
        make_material((0.9, 0.9, 0.9), (0.02, 0.02, 0.02), roughness=0.9, tex=(3, -1, -1)),  # 3 drape
        make_material((0.8, 0.8, 0.8), (0.1, 0.1, 0.1), roughness=0.6, tex=(0, 4, 5)),  # 4 columns
        make_material((0.7, 0.6, 0.2), (0.8, 0.7, 0.3), roughness=0.12),  # 5 brass
        make_material((0.9, 0.9, 0.9), roughness=0.95, tex=(7, -1, -1)),  # 6 plants
        make_material((0.9, 0.9, 0.9), (0.04, 0.04, 0.04), roughness=0.05, dissolve=0.0),  # 7 glass lamps
    ]
    mb = MeshBuilder()
    L, Wd, Ht = 30.0, 12.0, 14.0
    x0, x1, z0, z1 = -L / 2, L / 2, -Wd / 2, Wd / 2
    mb.quad_grid((x0, 0, z0), (0, 0, Wd), (L, 0, 0), q(60), q(120), 1, uv_scale=12)  # floor
    mb.quad_grid((x0, Ht, z0), (L, 0, 0), (0, 0, Wd), q(40), q(20), 0, uv_scale=8)  # roof
    mb.quad_grid((x0, 0, z0), (L, 0, 0), (0, Ht, 0), q(100), q(48), 0, uv_scale=10)
    mb.quad_grid((x0, 0, z1), (0, Ht, 0), (L, 0, 0), q(48), q(100), 0, uv_scale=10)
    mb.quad_grid((x0, 0, z0), (0, Ht, 0), (0, 0, Wd), q(48), q(40), 0, uv_scale=6)
    mb.quad_grid((x1, 0, z0), (0, 0, Wd), (0, Ht, 0), q(40), q(48), 0, uv_scale=6)
    # upper gallery floors along both long sides
    for zc in (z0 + 1.5, z1 - 1.5):
        mb.box((x0, 5.0, zc - 1.5), (x1, 5.3, zc + 1.5), 0, n=q(24))
    # two storeys of columns with capitals, arches between them
    for storey, (ybase, hcol) in enumerate(((0.0, 5.0), (5.3, 4.2))):
        for zc in (z0 + 3.0, z1 - 3.0):
            xs = np.linspace(x0 + 2.0, x1 - 2.0, 11)
            for i, xc in enumerate(xs):
                mb.cylinder((xc, ybase, zc), 0.32, hcol, q(40), 4, rings=q(16))
                mb.box((xc - 0.45, ybase + hcol - 0.35, zc - 0.45), (xc + 0.45, ybase + hcol, zc + 0.45), 4, n=q(3))
                if i + 1 < len(xs):  # arch: half torus of small segments
                    xa, xb = xc, xs[i + 1]
                    r = 0.5 * (xb - xa)
                    th = np.linspace(0.0, np.pi, q(24) + 1)
                    for t0, t1 in zip(th[:-1], th[1:]):
                        pa = ((xa + xb) / 2 - r * np.cos(t0), ybase + hcol - 0.9 + 0.9 * np.sin(t0))
                        pb = ((xa + xb) / 2 - r * np.cos(t1), ybase + hcol - 0.9 + 0.9 * np.sin(t1))
                        lo = (min(pa[0], pb[0]), min(pa[1], pb[1]) - 0.12, zc - 0.25)
                        hi = (max(pa[0], pb[0]) + 1e-3, max(pa[1], pb[1]) + 0.12, zc + 0.25)
                        mb.box(lo, hi, 0, n=1)
    # drapes hanging in the nave
    for k, xc in enumerate(np.linspace(x0 + 4.0, x1 - 4.0, 6)):
        ph = rng.uniform(0, 6.28)

        def wave(S, T, ph=ph):
            return np.stack([0.25 * np.sin(9.0 * T + ph) * S, np.zeros_like(S), 0.15 * np.sin(7.0 * S + ph)], -1)

        mb.quad_grid((xc, 9.5, -2.2), (0, -6.0, 0), (0, 0, 4.4), q(70), q(60), 2 + (k & 1), uv_scale=3, displace=wave)
    # lamps and plants
    for xc in np.linspace(x0 + 3.0, x1 - 3.0, 8):
        mb.sphere((xc, 4.2, 0.0), 0.22, q(32), q(16), 7)
        mb.cylinder((xc, 4.4, 0.0), 0.02, 5.0, q(8), 5, rings=1)
    for xc in np.linspace(x0 + 2.5, x1 - 2.5, 6):
        for zc in (z0 + 4.4, z1 - 4.4):
            mb.cylinder((xc, 0.0, zc), 0.3, 0.5, q(24), 5, rings=q(2))
            for _ in range(q(30)):
                off = rng.normal(0, 0.3, 3)
                mb.sphere((xc + off[0], 0.9 + abs(off[1]) * 1.4, zc + off[2]), 0.09, q(12), q(6), 6)
    sc = mb.finish(mats, tex)
    sc["name"] = "sponza_scale"
    sc["camera"] = _lookat_pose((x0 + 2.5, 2.2, 0.6), (x1, 5.0, -0.5))
    sc["light"] = {"pos": (0.0, Ht - 0.5, 0.0), "dir": (0.0, -1.0, 0.0)}
    return sc


def san_miguel_scale(seed=5, million_tris=10.0):
    """BASELINE.json configs[4] stand-in: a courtyard with arcades and dense vegetation of San-Miguel scale
    (~10 M triangles by default): tens of thousands of small leaves and finely tessellated architecture."""
    rng = np.random.default_rng(seed)
    target = int(million_tris * 1e6)
    mats = [
        make_material((0.72, 0.68, 0.6), roughness=0.9),  # 0 plaster
        make_material((0.5, 0.42, 0.36), (0.1, 0.1, 0.1), roughness=0.5),  # 1 paving
        make_material((0.2, 0.45, 0.12), (0.05, 0.05, 0.05), roughness=0.6),  # 2 leaves
        make_material((0.35, 0.25, 0.15), roughness=0.8),  # 3 wood
        make_material((0.9, 0.9, 0.9), (0.04, 0.04, 0.04), roughness=0.05, dissolve=0.0),  # 4 glass
        make_material((0.6, 0.6, 0.62), (0.7, 0.7, 0.7), roughness=0.2),  # 5 metal
    ]
    mb = MeshBuilder()
    S, Ht = 24.0, 9.0
    # architecture: ~15 % of the budget in finely tessellated planes
    g = max(8, int(np.sqrt(target * 0.15 / (2 * 6))))
    bump = lambda S_, T_: np.stack([np.zeros_like(S_), 0.01 * np.sin(40 * S_) * np.sin(40 * T_), np.zeros_like(S_)], -1)
    mb.quad_grid((-S / 2, 0, -S / 2), (0, 0, S), (S, 0, 0), g, g, 1, uv_scale=16, displace=bump)
    for (o, eu, ev) in (((-S / 2, 0, -S / 2), (S, 0, 0), (0, Ht, 0)), ((-S / 2, 0, S / 2), (0, Ht, 0), (S, 0, 0)),
                        ((-S / 2, 0, -S / 2), (0, Ht, 0), (0, 0, S)), ((S / 2, 0, -S / 2), (0, 0, S), (0, Ht, 0))):
        mb.quad_grid(o, eu, ev, g, g, 0, uv_scale=8)
    mb.quad_grid((-S / 2, Ht, -S / 2), (S, 0, 0), (0, 0, S), g, g, 0)
    for xc in np.linspace(-S / 2 + 3, S / 2 - 3, 7):
        for zc in (-S / 2 + 3.5, S / 2 - 3.5):
            mb.cylinder((xc, 0.0, zc), 0.3, 4.5, 96, 0, rings=64)
    # tables, chairs: a few thousand boxes
    for _ in range(40):
        c = rng.uniform(-S / 2 + 5, S / 2 - 5, 2)
        mb.box((c[0] - 0.6, 0.7, c[1] - 0.6), (c[0] + 0.6, 0.76, c[1] + 0.6), 3, n=12, rot_y=rng.uniform(0, 3.1))
        mb.sphere((c[0], 0.9, c[1]), 0.1, 48, 24, (4, 5)[_ & 1])
    built = sum(p.shape[0] for p in mb.P)
    # vegetation: leaves (2 triangles each) in clusters, fills the remaining budget
    nleaf = max(0, (target - built) // 2)
    ntree = 60
    centers = np.stack([rng.uniform(-S / 2 + 2, S / 2 - 2, ntree), rng.uniform(2.0, 6.5, ntree),
                        rng.uniform(-S / 2 + 2, S / 2 - 2, ntree)], -1)
    which = rng.integers(0, ntree, nleaf)
    pos = centers[which] + rng.normal(0, 1.0, (nleaf, 3)) * np.array([1.2, 0.8, 1.2])
    pos[:, 1] = np.abs(pos[:, 1])
    u = rng.normal(0, 1, (nleaf, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    w = np.cross(u, rng.normal(0, 1, (nleaf, 3)))
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    sz = rng.uniform(0.02, 0.05, (nleaf, 1))
    a, b, c, d = pos - u * sz - w * sz * 0.5, pos + u * sz - w * sz * 0.5, pos + u * sz + w * sz * 0.5, pos - u * sz + w * sz * 0.5
    tri = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.float32)
    mb.add(tri, 2)
    sc = mb.finish(mats)
    sc["name"] = "san_miguel_scale"
    sc["camera"] = _lookat_pose((-S / 2 + 1.5, 1.8, -S / 2 + 1.5), (2.0, 3.0, 2.0))
    sc["light"] = {"pos": (0.0, Ht - 0.4, 0.0), "dir": (0.0, -1.0, 0.0)}
    return sc


# ----------------------------------------------------------------------------------------------
# two-level scenes (vkRT_TLAS.cpp:58-72: instances of bottom-level structures with transforms)
# ----------------------------------------------------------------------------------------------
def affine(scale=(1.0, 1.0, 1.0), axis=(0.0, 1.0, 0.0), angle=0.0, translate=(0.0, 0.0, 0.0), shear_xy=0.0):
    """Column-major mat4 (16 floats, glm layout) of T * R(axis, angle) * Shear * S."""
    ax = np.asarray(axis, np.float64)
    ax = ax / np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)
    Sh = np.eye(3)
    Sh[0, 1] = shear_xy
    A = R @ Sh @ np.diag(np.broadcast_to(np.asarray(scale, np.float64), 3))
    M = np.eye(4)
    M[:3, :3] = A
    M[:3, 3] = translate
    return np.ascontiguousarray(M.T.reshape(16).astype(np.float32))  # column-major


def instanced_scene(seed=11, n_instances=24, detail=1.0, wild=False):
    """A closed room (structure 0, one identity instance) furnished with instances of three small models: a smooth
    sphere-on-a-stem "tree", a box and a bumpy quad-grid "shrub". Transforms: rotations about random axes, uniform and
    (wild=True) non-uniform scales, mirrors and shears. Returns a scene dict with two more keys, "blas" (BLAS_DTYPE)
    and "instances" (INSTANCE_DTYPE); vertices are in object space."""
    rng = np.random.default_rng(seed)
    mats = [
        make_material((0.7, 0.7, 0.68), roughness=0.9),  # 0 walls
        make_material((0.25, 0.5, 0.2), (0.05, 0.05, 0.05), roughness=0.5),  # 1 foliage
        make_material((0.45, 0.3, 0.2), roughness=0.8),  # 2 wood
        make_material((0.6, 0.6, 0.65), (0.6, 0.6, 0.6), roughness=0.15),  # 3 metal
        make_material((0.9, 0.9, 0.9), (0.04, 0.04, 0.04), roughness=0.05, dissolve=0.0),  # 4 glass
    ]
    S, H = 10.0, 4.0
    parts = []
    mb = MeshBuilder()  # structure 0: the room
    g = max(2, int(8 * detail))
    mb.quad_grid((-S / 2, 0, -S / 2), (0, 0, S), (S, 0, 0), g, g, 0)
    mb.quad_grid((-S / 2, H, -S / 2), (S, 0, 0), (0, 0, S), g, g, 0)
    for (o, eu, ev) in (((-S / 2, 0, -S / 2), (S, 0, 0), (0, H, 0)), ((-S / 2, 0, S / 2), (0, H, 0), (S, 0, 0)),
                        ((-S / 2, 0, -S / 2), (0, H, 0), (0, 0, S)), ((S / 2, 0, -S / 2), (0, 0, S), (0, H, 0))):
        mb.quad_grid(o, eu, ev, g, g, 0)
    parts.append(mb)
    mb = MeshBuilder()  # structure 1: tree
    mb.cylinder((0.0, 0.0, 0.0), 0.06, 0.7, max(6, int(10 * detail)), 2, rings=2)
    mb.sphere((0.0, 0.95, 0.0), 0.35, max(8, int(16 * detail)), max(6, int(10 * detail)), 1)
    parts.append(mb)
    mb = MeshBuilder()  # structure 2: box
    mb.box((-0.3, 0.0, -0.3), (0.3, 0.6, 0.3), 3, n=max(1, int(2 * detail)))
    parts.append(mb)
    mb = MeshBuilder()  # structure 3: shrub
    bump = lambda S_, T_: np.stack([np.zeros_like(S_), 0.15 * np.sin(9 * S_) * np.cos(7 * T_), np.zeros_like(S_)], -1)
    mb.quad_grid((-0.4, 0.3, -0.4), (0.8, 0, 0), (0, 0, 0.8), max(4, int(10 * detail)), max(4, int(10 * detail)), 4,
                 displace=bump)
    parts.append(mb)
    scs = [m.finish(mats) for m in parts]
    verts = np.concatenate([x["vertices"] for x in scs])
    ntris = [len(x["indices"]) // 3 for x in scs]
    blas = np.zeros(len(scs), BLAS_DTYPE)
    blas["n_triangles"] = ntris
    blas["first_triangle"] = np.concatenate([[0], np.cumsum(ntris)[:-1]])
    inst = np.zeros(1 + n_instances, INSTANCE_DTYPE)
    inst[0]["transform"] = affine()
    for i in range(1, 1 + n_instances):
        k = int(rng.integers(1, len(scs)))
        sc_u = float(rng.uniform(0.5, 1.6))
        scale = (sc_u, sc_u, sc_u)
        shear = 0.0
        axis = (0.0, 1.0, 0.0)
        if wild:
            scale = tuple(float(v) for v in sc_u * rng.uniform(0.4, 1.5, 3) * rng.choice([1.0, -1.0], 3, p=[0.8, 0.2]))
            shear = float(rng.uniform(-0.5, 0.5))
            axis = tuple(rng.normal(0, 1, 3))
        pos = (float(rng.uniform(-S / 2 + 1, S / 2 - 1)), float(rng.uniform(0.0, 1.5 if wild else 0.0)),
               float(rng.uniform(-S / 2 + 1, S / 2 - 1)))
        inst[i]["transform"] = affine(scale, axis, float(rng.uniform(0, 2 * np.pi)), pos, shear)
        inst[i]["blas"] = k
        inst[i]["instance_id"] = 100 + i
    out = {"vertices": verts, "indices": np.arange(len(verts), dtype=np.uint32), "materials": scs[0]["materials"],
           "textures": [], "blas": blas, "instances": inst, "name": "instanced"}
    out["camera"] = _lookat_pose((-S / 2 + 0.8, 1.7, -S / 2 + 0.8), (0.5, 0.8, 0.5))
    out["light"] = {"pos": (0.0, H - 0.3, 0.0), "dir": (0.0, -1.0, 0.0)}
    return out
