"""ctypes binding of include/prt.h (libprt_hip.so). Plumbing only: every call goes to the HIP
library; there is no Python or CPU fallback — a missing library or GPU raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRT_LIB") or os.path.join(_HERE, "libprt_hip.so")  # PRT_LIB: measurement builds only

MODE_PATH, MODE_AO = 0, 1
FLAG_LITERAL, FLAG_RUSSIAN_ROULETTE = 1, 2

BVH_NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("first", "<u4"), ("bmax", "<f4", 3), ("count", "<u4")])
RAY_DTYPE = np.dtype([("o", "<f4", 3), ("tmin", "<f4"), ("d", "<f4", 3), ("tmax", "<f4")])
HIT_DTYPE = np.dtype([("prim", "<u4"), ("u", "<f4"), ("v", "<f4"), ("t", "<f4")])
from .scenes import BLAS_DTYPE, INSTANCE_DTYPE  # include/prt_types.h: prt_blas_range, prt_instance  # noqa: E402

EXPORTS = [
    "prt_create", "prt_destroy", "prt_last_error", "prt_set_scene", "prt_build_accel", "prt_set_scramble_seed",
    "prt_upload_scrambles", "prt_set_uniforms", "prt_set_flags", "prt_render_iteration", "prt_render", "prt_sync",
    "prt_clear_accum", "prt_read_accum", "prt_device_accum_ptr", "prt_bind_accum", "prt_postprocess",
    "prt_read_hit_ids", "prt_get_stats", "prt_reset_stats", "prt_trace", "prt_trace_device", "prt_trace_device_noalloc", "prt_read_bvh",
    "prt_read_sobol", "prt_set_profiling", "prt_set_profiling_mask", "prt_get_profile", "prt_set_pipeline_depth", "prt_get_pipeline_depth",
    "prt_update_vertices", "prt_refit_accel", "prt_write_accum", "prt_set_iteration_batch", "prt_get_iteration_batch",
    "prt_set_pass_memory_limit", "prt_get_pass_iterations", "prt_debug_read_stats",
    "prt_group_create", "prt_group_destroy", "prt_group_last_error", "prt_group_size", "prt_group_uses_rccl",
    "prt_group_context", "prt_group_set_scene", "prt_group_set_scramble_seed", "prt_group_set_uniforms",
    "prt_group_set_flags", "prt_group_clear_accum", "prt_group_render", "prt_group_reduce", "prt_group_read_accum",
    "prt_group_get_stats", "prt_set_instances", "prt_update_instance_transforms", "prt_read_blas_bvh",
    "prt_group_set_instances",
]

K_GENERATE, K_EXTEND, K_SHADE, K_SHADOW, K_AO_SHADE, K_RESOLVE, K_SHADOW_EXTEND, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 8
KERNEL_NAMES = {K_GENERATE: "generate", K_EXTEND: "extend", K_SHADE: "shade", K_SHADOW: "shadow",
                K_AO_SHADE: "ao_shade", K_RESOLVE: "resolve", K_SHADOW_EXTEND: "shadow_extend"}


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("iterations", C.c_uint64), ("bvh_nodes", C.c_uint32),
                ("bvh_inner_nodes", C.c_uint32), ("bvh_depth", C.c_uint32), ("bvh_build_ms", C.c_float),
                ("bvh_refit_ms", C.c_float), ("trace_overruns", C.c_uint32)]


class WaveSteps(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("node_steps", "node_lanes", "tri_steps", "tri_lanes", "setup_steps",
                                          "setup_lanes", "setup_ticks", "node_ticks", "tri_ticks", "total_ticks")]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_uint64 * 8), ("closest_rays", C.c_uint64),
                ("any_rays", C.c_uint64), ("closest_nodes", C.c_uint64), ("closest_tris", C.c_uint64),
                ("any_nodes", C.c_uint64), ("any_tris", C.c_uint64), ("closest_steps", WaveSteps),
                ("any_steps", WaveSteps), ("fused_closest_rays", C.c_uint64), ("fused_any_rays", C.c_uint64)]


class _Tex(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class PrtError(RuntimeError):
    pass


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PrtError(f"{LIB_PATH} is missing: build it with `make -C pathtracer_amd/csrc` "
                           "(or __graft_entry__.build()); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        vp, u32, sz, i32 = C.c_void_p, C.c_uint32, C.c_size_t, C.c_int
        L.prt_create.argtypes = [i32, u32, u32, C.POINTER(vp)]
        L.prt_destroy.argtypes = [vp]
        L.prt_last_error.argtypes = [vp]
        L.prt_last_error.restype = C.c_char_p
        L.prt_set_scene.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz]
        L.prt_build_accel.argtypes = [vp]
        L.prt_set_scramble_seed.argtypes = [vp, u32]
        L.prt_upload_scrambles.argtypes = [vp, vp, u32]
        L.prt_set_uniforms.argtypes = [vp, vp]
        L.prt_set_flags.argtypes = [vp, u32]
        L.prt_render_iteration.argtypes = [vp, i32]
        L.prt_render.argtypes = [vp, i32, u32, u32, u32]
        L.prt_sync.argtypes = [vp]
        L.prt_clear_accum.argtypes = [vp]
        L.prt_read_accum.argtypes = [vp, vp]
        L.prt_device_accum_ptr.argtypes = [vp, C.POINTER(vp)]
        L.prt_bind_accum.argtypes = [vp, vp]
        L.prt_postprocess.argtypes = [vp, vp]
        L.prt_read_hit_ids.argtypes = [vp, vp]
        L.prt_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.prt_reset_stats.argtypes = [vp]
        L.prt_trace.argtypes = [vp, vp, sz, i32, vp]
        L.prt_trace_device.argtypes = [vp, vp, sz, i32, vp, C.POINTER(C.c_float)]
        L.prt_trace_device_noalloc.argtypes = [vp, vp, sz, i32, vp, vp]
        L.prt_read_bvh.argtypes = [vp, vp, vp]
        L.prt_read_sobol.argtypes = [vp, vp]
        L.prt_set_profiling.argtypes = [vp, i32]
        L.prt_get_profile.argtypes = [vp, C.POINTER(Profile)]
        L.prt_set_profiling_mask.argtypes = [vp, C.c_uint32]
        L.prt_set_pipeline_depth.argtypes = [vp, i32]
        L.prt_get_pipeline_depth.argtypes = [vp]
        L.prt_set_iteration_batch.argtypes = [vp, i32]
        L.prt_get_iteration_batch.argtypes = [vp]
        L.prt_set_pass_memory_limit.argtypes = [vp, C.c_uint64]
        L.prt_get_pass_iterations.argtypes = [vp]
        L.prt_update_vertices.argtypes = [vp, vp, sz]
        L.prt_refit_accel.argtypes = [vp]
        L.prt_write_accum.argtypes = [vp, vp]
        L.prt_set_instances.argtypes = [vp, vp, sz, vp, sz]
        L.prt_update_instance_transforms.argtypes = [vp, vp, sz]
        L.prt_read_blas_bvh.argtypes = [vp, u32, vp, vp, C.POINTER(u32)]
        L.prt_group_create.argtypes = [i32, C.POINTER(i32), u32, u32, C.POINTER(vp)]
        L.prt_group_destroy.argtypes = [vp]
        L.prt_group_last_error.argtypes = [vp]
        L.prt_group_last_error.restype = C.c_char_p
        L.prt_group_size.argtypes = [vp]
        L.prt_group_uses_rccl.argtypes = [vp]
        L.prt_group_context.argtypes = [vp, i32]
        L.prt_group_context.restype = vp
        L.prt_group_set_scene.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz]
        L.prt_group_set_instances.argtypes = [vp, vp, sz, vp, sz]
        L.prt_group_set_scramble_seed.argtypes = [vp, u32]
        L.prt_group_set_uniforms.argtypes = [vp, vp]
        L.prt_group_set_flags.argtypes = [vp, u32]
        L.prt_group_clear_accum.argtypes = [vp]
        L.prt_group_render.argtypes = [vp, i32, u32, u32]
        L.prt_group_reduce.argtypes = [vp, i32]
        L.prt_group_read_accum.argtypes = [vp, i32, vp]
        L.prt_group_get_stats.argtypes = [vp, C.POINTER(Stats)]
        for name in EXPORTS:
            if name not in ("prt_last_error", "prt_group_last_error", "prt_group_context"):
                getattr(L, name).restype = i32
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One prt_context == one VkRTX instance (reference: src/vkRTX_setup.h:17-40) on one GPU."""

    def __init__(self, width, height, device=0):
        self.L = lib()
        self.w, self.h = int(width), int(height)
        h = C.c_void_p()
        rc = self.L.prt_create(device, self.w, self.h, C.byref(h))
        if rc != 0:
            raise PrtError(f"prt_create: {self.L.prt_last_error(None).decode()} (code {rc})")
        self.h_ = h
        self.ntri = 0

    def _chk(self, rc):
        if rc != 0:
            raise PrtError(f"{self.L.prt_last_error(self.h_).decode()} (code {rc})")

    def close(self):
        if getattr(self, "h_", None):
            self.L.prt_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- scene -------------------------------------------------------------------------------
    def set_scene(self, scene):
        v = np.ascontiguousarray(scene["vertices"])
        i = np.ascontiguousarray(scene["indices"], dtype=np.uint32)
        m = np.ascontiguousarray(scene["materials"])
        texs = [np.ascontiguousarray(t, dtype=np.uint8) for t in scene.get("textures", [])]
        arr = (_Tex * max(1, len(texs)))()
        for k, t in enumerate(texs):
            arr[k].rgba = t.ctypes.data
            arr[k].height, arr[k].width = t.shape[0], t.shape[1]
        self._chk(self.L.prt_set_scene(self.h_, _ptr(v), len(v), _ptr(i), len(i), _ptr(m), len(m),
                                       C.addressof(arr), len(texs)))
        self.ntri = len(i) // 3

    def set_instances(self, blas, instances):
        """Two-level scene (call between set_scene and build_accel): blas = BLAS_DTYPE array of triangle ranges,
        instances = INSTANCE_DTYPE array. Empty instances: back to the single-level scene."""
        b = np.ascontiguousarray(blas, dtype=BLAS_DTYPE)
        i = np.ascontiguousarray(instances, dtype=INSTANCE_DTYPE)
        self._chk(self.L.prt_set_instances(self.h_, _ptr(b), len(b), _ptr(i), len(i)))
        self._blas = b

    def update_instance_transforms(self, transforms):
        t = np.ascontiguousarray(transforms, dtype=np.float32).reshape(-1, 16)
        self._chk(self.L.prt_update_instance_transforms(self.h_, _ptr(t), len(t)))

    def read_blas_bvh(self, k):
        n = C.c_uint32()
        self._chk(self.L.prt_read_blas_bvh(self.h_, k, None, None, C.byref(n)))
        nodes = np.zeros(n.value, BVH_NODE_DTYPE)
        order = np.zeros(int(self._blas[k]["n_triangles"]), np.uint32)
        self._chk(self.L.prt_read_blas_bvh(self.h_, k, _ptr(nodes), _ptr(order), None))
        return nodes, order

    def build_accel(self):
        self._chk(self.L.prt_build_accel(self.h_))

    def update_vertices(self, vertices):
        v = np.ascontiguousarray(vertices)
        self._chk(self.L.prt_update_vertices(self.h_, _ptr(v), len(v)))

    def refit_accel(self):
        self._chk(self.L.prt_refit_accel(self.h_))

    def set_seed(self, seed):
        self._chk(self.L.prt_set_scramble_seed(self.h_, seed))

    def upload_scrambles(self, table):
        table = np.ascontiguousarray(table, dtype=np.uint32)
        assert table.size == 32 * self.w * self.h
        self._chk(self.L.prt_upload_scrambles(self.h_, _ptr(table), 32))

    def set_uniforms(self, ubo):
        ubo = np.ascontiguousarray(ubo)
        assert ubo.nbytes == 464
        self._chk(self.L.prt_set_uniforms(self.h_, _ptr(ubo)))

    def set_flags(self, flags):
        self._chk(self.L.prt_set_flags(self.h_, flags))

    # -- render ------------------------------------------------------------------------------
    def render_iteration(self, mode):
        self._chk(self.L.prt_render_iteration(self.h_, mode))

    def render(self, mode, first_iter, n_iter, stride=1):
        self._chk(self.L.prt_render(self.h_, mode, first_iter, n_iter, stride))

    def sync(self):
        self._chk(self.L.prt_sync(self.h_))

    def set_pipeline_depth(self, n):
        self._chk(self.L.prt_set_pipeline_depth(self.h_, n))

    def pipeline_depth(self):
        return int(self.L.prt_get_pipeline_depth(self.h_))

    def set_iteration_batch(self, n):
        self._chk(self.L.prt_set_iteration_batch(self.h_, n))

    def iteration_batch(self):
        return int(self.L.prt_get_iteration_batch(self.h_))

    def set_pass_memory_limit(self, nbytes):
        """Upper bound of the per-pass buffers of prt_render (0 = none): larger passes are split (same image)."""
        self._chk(self.L.prt_set_pass_memory_limit(self.h_, int(nbytes)))

    def pass_iterations(self):
        """Iterations per pass the last render() actually used (the memory policy may have shrunk the batch)."""
        return int(self.L.prt_get_pass_iterations(self.h_))

    def clear_accum(self):
        self._chk(self.L.prt_clear_accum(self.h_))

    def read_accum(self):
        out = np.empty((self.h, self.w, 4), np.float32)
        self._chk(self.L.prt_read_accum(self.h_, _ptr(out)))
        return out

    def write_accum(self, rgba):
        a = np.ascontiguousarray(rgba, dtype=np.float32)
        assert a.size == self.w * self.h * 4
        self._chk(self.L.prt_write_accum(self.h_, _ptr(a)))

    def device_accum_ptr(self):
        p = C.c_void_p()
        self._chk(self.L.prt_device_accum_ptr(self.h_, C.byref(p)))
        return p.value

    def bind_accum(self, dev_ptr):
        self._chk(self.L.prt_bind_accum(self.h_, C.c_void_p(dev_ptr)))

    def postprocess(self):
        out = np.empty((self.h, self.w, 4), np.uint8)
        self._chk(self.L.prt_postprocess(self.h_, _ptr(out)))
        return out

    def read_hit_ids(self):
        out = np.empty((self.h, self.w), np.uint32)
        self._chk(self.L.prt_read_hit_ids(self.h_, _ptr(out)))
        return out

    def stats(self):
        s = Stats()
        self._chk(self.L.prt_get_stats(self.h_, C.byref(s)))
        return s

    def reset_stats(self):
        self._chk(self.L.prt_reset_stats(self.h_))

    def set_profiling(self, level):
        self._chk(self.L.prt_set_profiling(self.h_, level))

    def set_profiling_mask(self, kinds):
        self._chk(self.L.prt_set_profiling_mask(self.h_, kinds))

    def profile(self):
        p = Profile()
        self._chk(self.L.prt_get_profile(self.h_, C.byref(p)))
        return p

    # -- inspection --------------------------------------------------------------------------
    def trace(self, rays, any_hit=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(len(rays), HIT_DTYPE)
        self._chk(self.L.prt_trace(self.h_, _ptr(rays), len(rays), 1 if any_hit else 0, _ptr(out)))
        return out

    def trace_device(self, rays_ptr, n, hits_ptr, any_hit=False):
        ms = C.c_float()
        self._chk(self.L.prt_trace_device(self.h_, C.c_void_p(rays_ptr), n, 1 if any_hit else 0,
                                          C.c_void_p(hits_ptr), C.byref(ms)))
        return ms.value

    def trace_device_noalloc(self, rays_ptr, n, hits_ptr, any_hit=False, stream=None):
        self._chk(self.L.prt_trace_device_noalloc(self.h_, C.c_void_p(rays_ptr), n, 1 if any_hit else 0,
                                                  C.c_void_p(hits_ptr), C.c_void_p(stream)))

    def read_bvh(self):
        s = self.stats()
        nodes = np.zeros(s.bvh_nodes, BVH_NODE_DTYPE)
        order = np.zeros(self.ntri, np.uint32)
        self._chk(self.L.prt_read_bvh(self.h_, _ptr(nodes), _ptr(order)))
        return nodes, order

    def read_sobol(self):
        out = np.zeros((64, 32), np.uint32)
        self._chk(self.L.prt_read_sobol(self.h_, _ptr(out)))
        return out


class Group:
    """prt_group: several GPUs (or several contexts on one GPU) behind one handle, sample-sharded, one reduce."""

    def __init__(self, ordinals, width, height):
        self.L = lib()
        self.w, self.h = int(width), int(height)
        arr = (C.c_int * len(ordinals))(*ordinals)
        h = C.c_void_p()
        rc = self.L.prt_group_create(len(ordinals), arr, self.w, self.h, C.byref(h))
        if rc != 0:
            raise PrtError(f"prt_group_create: {self.L.prt_group_last_error(None).decode()} (code {rc})")
        self.h_ = h

    def _chk(self, rc):
        if rc != 0:
            raise PrtError(f"{self.L.prt_group_last_error(self.h_).decode()} (code {rc})")

    def close(self):
        if getattr(self, "h_", None):
            self.L.prt_group_destroy(self.h_)
            self.h_ = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return int(self.L.prt_group_size(self.h_))

    def uses_rccl(self):
        return bool(self.L.prt_group_uses_rccl(self.h_))

    def set_scene(self, scene):
        v = np.ascontiguousarray(scene["vertices"])
        i = np.ascontiguousarray(scene["indices"], dtype=np.uint32)
        m = np.ascontiguousarray(scene["materials"])
        texs = [np.ascontiguousarray(t, dtype=np.uint8) for t in scene.get("textures", [])]
        arr = (_Tex * max(1, len(texs)))()
        for k, t in enumerate(texs):
            arr[k].rgba = t.ctypes.data
            arr[k].height, arr[k].width = t.shape[0], t.shape[1]
        self._chk(self.L.prt_group_set_scene(self.h_, _ptr(v), len(v), _ptr(i), len(i), _ptr(m), len(m),
                                             C.addressof(arr), len(texs)))

    def set_instances(self, blas, instances):
        b = np.ascontiguousarray(blas, dtype=BLAS_DTYPE)
        i = np.ascontiguousarray(instances, dtype=INSTANCE_DTYPE)
        self._chk(self.L.prt_group_set_instances(self.h_, _ptr(b), len(b), _ptr(i), len(i)))

    def set_seed(self, seed):
        self._chk(self.L.prt_group_set_scramble_seed(self.h_, seed))

    def set_uniforms(self, ubo):
        ubo = np.ascontiguousarray(ubo)
        assert ubo.nbytes == 464
        self._chk(self.L.prt_group_set_uniforms(self.h_, _ptr(ubo)))

    def set_flags(self, flags):
        self._chk(self.L.prt_group_set_flags(self.h_, flags))

    def clear_accum(self):
        self._chk(self.L.prt_group_clear_accum(self.h_))

    def render(self, mode, first_iter, n_iter):
        self._chk(self.L.prt_group_render(self.h_, mode, first_iter, n_iter))

    def reduce(self, root=0):
        self._chk(self.L.prt_group_reduce(self.h_, root))

    def read_accum(self, root=0):
        out = np.empty((self.h, self.w, 4), np.float32)
        self._chk(self.L.prt_group_read_accum(self.h_, root, _ptr(out)))
        return out

    def stats(self):
        s = Stats()
        self._chk(self.L.prt_group_get_stats(self.h_, C.byref(s)))
        return s

    def member_accum(self, member):
        """prt_read_accum on one member's context (prt_group_context)."""
        c = self.L.prt_group_context(self.h_, member)
        if not c:
            raise PrtError("prt_group_context: no such member")
        out = np.empty((self.h, self.w, 4), np.float32)
        if self.L.prt_read_accum(C.c_void_p(c), _ptr(out)) != 0:
            raise PrtError(self.L.prt_last_error(C.c_void_p(c)).decode())
        return out
