"""ctypes wrapper over oracle/liboracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product (pathtracer_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("first", "<u4"), ("bmax", "<f4", 3), ("count", "<u4")])
HIT_DTYPE = np.dtype([("prim", "<u4"), ("u", "<f4"), ("v", "<f4"), ("t", "<f4")])
RAY_DTYPE = np.dtype([("o", "<f4", 3), ("tmin", "<f4"), ("d", "<f4", 3), ("tmax", "<f4")])

BLAS_DTYPE = np.dtype([("first_triangle", "<u4"), ("n_triangles", "<u4")])
INSTANCE_DTYPE = np.dtype([("transform", "<f4", 16), ("blas", "<u4"), ("instance_id", "<u4"), ("hit_group", "<u4"),
                           ("pad", "<u4")])

FLAG_LITERAL = 1
FLAG_RUSSIAN_ROULETTE = 2
FLAG_BRUTE_FORCE = 4


class _Tex(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.cpp")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, u32, sz, i32, f32 = C.c_void_p, C.c_uint32, C.c_size_t, C.c_int, C.c_float
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [vp, sz, vp, sz, vp, sz, vp, sz]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_set_sobol.argtypes = [vp, vp, u32, u32]
        L.orc_set_scramble_seed.argtypes = [vp, u32]
        L.orc_set_scramble_table.argtypes = [vp, vp, u32, u32]
        L.orc_build_bvh.argtypes = [vp]
        L.orc_build_bvh.restype = i32
        L.orc_bvh_num_nodes.argtypes = [vp]
        L.orc_bvh_num_nodes.restype = u32
        L.orc_bvh_depth.argtypes = [vp]
        L.orc_bvh_depth.restype = u32
        L.orc_bvh_read.argtypes = [vp, vp, vp]
        L.orc_sobol_sample.argtypes = [vp, u32, u32, u32]
        L.orc_sobol_sample.restype = f32
        L.orc_scramble.argtypes = [vp, u32, u32, u32, u32, u32]
        L.orc_scramble.restype = u32
        L.orc_sincos.argtypes = [f32, C.POINTER(f32), C.POINTER(f32)]
        L.orc_primary_ray.argtypes = [vp, u32, u32, u32, u32, f32, f32, vp]
        L.orc_trace_closest.argtypes = [vp, vp, sz, vp, u32, i32]
        L.orc_trace_any.argtypes = [vp, vp, sz, vp, u32, i32]
        L.orc_render.argtypes = [vp, vp, u32, u32, i32, u32, vp, vp, i32]
        L.orc_render.restype = C.c_uint64
        L.orc_postprocess.argtypes = [vp, u32, u32, vp]
        L.orc_set_instances.argtypes = [vp, vp, sz, vp, sz]
        L.orc_set_instances.restype = i32
        L.orc_inverse4.argtypes = [vp, vp]
        L.orc_blas_num_nodes.argtypes = [vp, u32]
        L.orc_blas_num_nodes.restype = u32
        L.orc_blas_read.argtypes = [vp, u32, vp, vp]
        _LIB = L
    return _LIB


def golden_sobol():
    p = os.path.join(_HERE, "..", "tests", "golden", "sobol_64x52.u32")
    return np.fromfile(p, dtype="<u4").reshape(64, 52)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, scene, seed=1, sobol=None):
        L = lib()
        self._keep = []
        v = np.ascontiguousarray(scene["vertices"])
        i = np.ascontiguousarray(scene["indices"], dtype=np.uint32)
        m = np.ascontiguousarray(scene["materials"])
        texs = scene.get("textures", [])
        arr = (_Tex * max(1, len(texs)))()
        for k, t in enumerate(texs):
            t = np.ascontiguousarray(t, dtype=np.uint8)
            self._keep.append(t)
            arr[k].rgba = t.ctypes.data
            arr[k].height, arr[k].width = t.shape[0], t.shape[1]
        self.h = L.orc_scene_create(_ptr(v), len(v), _ptr(i), len(i), _ptr(m), len(m), C.addressof(arr), len(texs))
        self.ntri = len(i) // 3
        sob = golden_sobol() if sobol is None else sobol
        sob = np.ascontiguousarray(sob, dtype=np.uint32)
        L.orc_set_sobol(self.h, _ptr(sob), sob.shape[0], sob.shape[1])
        L.orc_set_scramble_seed(self.h, seed)
        self.built = False

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_seed(self, seed):
        lib().orc_set_scramble_seed(self.h, seed)

    def set_scramble_table(self, table, w, h):
        table = np.ascontiguousarray(table, dtype=np.uint32)
        self._keep.append(table)
        lib().orc_set_scramble_table(self.h, _ptr(table), w, h)

    def set_instances(self, blas, instances):
        """Two-level scene: blas = BLAS_DTYPE array (triangle ranges), instances = INSTANCE_DTYPE array
        (transform = column-major mat4 object -> world). Empty instances: back to the single-level scene."""
        b = np.ascontiguousarray(blas, dtype=BLAS_DTYPE)
        i = np.ascontiguousarray(instances, dtype=INSTANCE_DTYPE)
        if lib().orc_set_instances(self.h, _ptr(b), len(b), _ptr(i), len(i)) != 0:
            raise RuntimeError("orc_set_instances: bad range or index")
        self._blas = b
        self.built = False

    def blas_bvh(self, k):
        n = lib().orc_blas_num_nodes(self.h, k)
        nodes = np.zeros(n, NODE_DTYPE)
        order = np.zeros(int(self._blas[k]["n_triangles"]), np.uint32)
        lib().orc_blas_read(self.h, k, _ptr(nodes), _ptr(order))
        return nodes, order

    def build_bvh(self):
        rc = lib().orc_build_bvh(self.h)
        if rc != 0:
            raise RuntimeError("orc_build_bvh failed")
        self.built = True

    def bvh(self):
        n = lib().orc_bvh_num_nodes(self.h)
        nodes = np.zeros(n, NODE_DTYPE)
        order = np.zeros(self.ntri, np.uint32)
        lib().orc_bvh_read(self.h, _ptr(nodes), _ptr(order))
        return nodes, order

    def bvh_depth(self):
        return lib().orc_bvh_depth(self.h)

    def sobol(self, index, dim, scramble):
        return lib().orc_sobol_sample(self.h, index, dim, scramble)

    def scramble(self, px, py, w, h, layer):
        return lib().orc_scramble(self.h, px, py, w, h, layer)

    def trace_closest(self, rays, brute=False, threads=0):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(len(rays), HIT_DTYPE)
        if not brute and not self.built:
            self.build_bvh()
        lib().orc_trace_closest(self.h, _ptr(rays), len(rays), _ptr(out), FLAG_BRUTE_FORCE if brute else 0, threads)
        return out

    def trace_any(self, rays, brute=False, threads=0):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(len(rays), np.uint8)
        if not brute and not self.built:
            self.build_bvh()
        lib().orc_trace_any(self.h, _ptr(rays), len(rays), _ptr(out), FLAG_BRUTE_FORCE if brute else 0, threads)
        return out

    def render(self, ubo, w, h, mode, accum=None, flags=0, threads=0, want_ids=False):
        if not (flags & FLAG_BRUTE_FORCE) and not self.built:
            self.build_bvh()
        if accum is None:
            accum = np.zeros((h, w, 4), np.float32)
        assert accum.dtype == np.float32 and accum.flags["C_CONTIGUOUS"] and accum.size == w * h * 4
        ids = np.zeros((h, w), np.uint32) if want_ids else None
        ubo = np.ascontiguousarray(ubo)
        rays = lib().orc_render(self.h, _ptr(ubo), w, h, mode, flags, _ptr(accum), _ptr(ids) if want_ids else None,
                                threads)
        return accum, ids, int(rays)


def primary_ray(ubo, px, py, w, h, varx=0.5, vary=0.5):
    out = np.zeros((), RAY_DTYPE)
    ubo = np.ascontiguousarray(ubo)
    lib().orc_primary_ray(_ptr(ubo), px, py, w, h, varx, vary, _ptr(out))
    return out


def inverse4(m):
    a = np.ascontiguousarray(m, dtype=np.float32).reshape(16)
    out = np.zeros(16, np.float32)
    lib().orc_inverse4(_ptr(a), _ptr(out))
    return out


def sincos(x):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def postprocess(accum, w, h):
    out = np.zeros((h, w, 4), np.uint8)
    a = np.ascontiguousarray(accum, dtype=np.float32)
    lib().orc_postprocess(_ptr(a), w, h, _ptr(out))
    return out


# ---- the shading helpers one at a time (oracle.h: orc_helper) -------------------------------------------------------
(H_FORM_BASIS, H_SAMPLE_LIGHT, H_GGX_DISTRIBUTION, H_GGX_G1, H_FRESNEL_DIELECTRIC, H_SCHLICK_FRESNEL, H_SMITH_MASKING,
 H_SMITH_MASKING_SHADOWING, H_SAMPLE_VNDF, H_REFRACT, H_PRIMARY_RAY, H_MITCHELL, H_MITCHELL_WEIGHT) = range(13)
HELPER_IO = {H_FORM_BASIS: (3, 9), H_SAMPLE_LIGHT: (20, 4), H_GGX_DISTRIBUTION: (4, 1), H_GGX_G1: (7, 1),
             H_FRESNEL_DIELECTRIC: (3, 1), H_SCHLICK_FRESNEL: (4, 3), H_SMITH_MASKING: (7, 1),
             H_SMITH_MASKING_SHADOWING: (7, 1), H_SAMPLE_VNDF: (6, 3), H_REFRACT: (7, 4), H_PRIMARY_RAY: (22, 6),
             H_MITCHELL: (1, 1), H_MITCHELL_WEIGHT: (2, 1)}


def helper(which, inputs, fn=None):
    """Rows of `inputs` (n x n_in float32) through helper `which`; returns n x n_out float32. `fn`: another library's
    function with orc_helper's signature (the transcription audit)."""
    if fn is None:
        fn = lib().orc_helper
    fn.argtypes, fn.restype = [C.c_int, C.c_void_p, C.c_void_p], C.c_int
    n_in, n_out = HELPER_IO[which]
    a = np.ascontiguousarray(np.atleast_2d(np.asarray(inputs, np.float32)))
    assert a.shape[1] == n_in
    out = np.zeros((len(a), n_out), np.float32)
    for k in range(len(a)):
        if fn(which, a[k].ctypes.data, out[k].ctypes.data) != n_out:
            raise RuntimeError(f"helper {which}")
    return out
