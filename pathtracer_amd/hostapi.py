"""ctypes binding of include/prt_host.h (libprt_host.so): the C++ host layer — Model / CameraControls /
AreaLight / Renderer of the reference's scene API. Plumbing for tests and tools; no fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import scenes, uniforms

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libprt_host.so")

EXPORTS = [
    "prth_model_load", "prth_model_free", "prth_model_num_vertices", "prth_model_num_indices",
    "prth_model_num_materials", "prth_model_num_textures", "prth_model_vertices", "prth_model_indices",
    "prth_model_materials", "prth_model_texture", "prth_model_warnings", "prth_obj_dump_json", "prth_image_load",
    "prth_free", "prth_default_settings", "prth_camera_uniforms", "prth_render_obj", "prth_flatten_instances",
    "prth_render_model", "prth_write_png", "prth_checkpoint_write", "prth_checkpoint_read", "prth_frame_loop",
    "prth_pool_instances", "prth_render_instances",
]

EVENT_MOVE_CAMERA, EVENT_MOVE_LIGHT, EVENT_SET_MODE = 0, 1, 2


class Settings(C.Structure):
    _fields_ = [("numIndicesBounces", C.c_int32), ("samplesPerPixel", C.c_int32), ("numAArays", C.c_int32),
                ("filterRadius", C.c_float), ("numAOrays", C.c_int32), ("aoRayLength", C.c_float),
                ("lightSourceArea", C.c_float), ("lightE", C.c_float), ("lightOtherE", C.c_float),
                ("rtRenderingMode", C.c_int32), ("iteration", C.c_uint32)]


class _Tex(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class _Instance(C.Structure):
    _fields_ = [("modelIndex", C.c_uint32), ("transform", C.c_float * 16), ("instanceId", C.c_uint32),
                ("hitGroupIndex", C.c_uint32)]


class HostError(RuntimeError):
    pass


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HostError(f"{LIB_PATH} is missing: build it with `make -C pathtracer_amd/host`")
        L = C.CDLL(LIB_PATH)
        vp, sz = C.c_void_p, C.c_size_t
        L.prth_model_load.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, sz]
        L.prth_model_free.argtypes = [vp]
        L.prth_model_free.restype = None
        for n in ("vertices", "indices", "materials", "textures"):
            f = getattr(L, "prth_model_num_" + n)
            f.argtypes, f.restype = [vp], sz
        for n in ("vertices", "indices", "materials"):
            f = getattr(L, "prth_model_" + n)
            f.argtypes, f.restype = [vp], vp
        L.prth_model_texture.argtypes = [vp, sz, C.POINTER(_Tex), C.POINTER(C.c_char_p)]
        L.prth_model_warnings.argtypes = [vp]
        L.prth_model_warnings.restype = C.c_char_p
        L.prth_obj_dump_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        L.prth_image_load.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(vp)]
        L.prth_free.argtypes = [vp]
        L.prth_free.restype = None
        L.prth_default_settings.argtypes = [C.POINTER(Settings)]
        L.prth_default_settings.restype = None
        fp = C.POINTER(C.c_float)
        L.prth_camera_uniforms.argtypes = [C.c_float] * 4 + [fp, C.c_float, C.c_float, C.POINTER(Settings), vp, fp]
        L.prth_render_obj.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(Settings), fp, fp, fp, fp,
                                      C.c_uint32, C.c_uint32, vp, vp, C.c_char_p, sz]
        L.prth_flatten_instances.argtypes = [C.POINTER(vp), sz, C.POINTER(_Instance), sz, C.POINTER(vp),
                                             C.POINTER(C.c_uint32), C.c_char_p, sz]
        L.prth_pool_instances.argtypes = [C.POINTER(vp), sz, C.POINTER(_Instance), sz, C.POINTER(vp), vp, vp, C.c_char_p, sz]
        L.prth_render_instances.argtypes = [C.POINTER(vp), sz, C.POINTER(_Instance), sz, C.c_int, C.c_int, C.c_uint32,
                                            C.c_uint32, C.POINTER(Settings), fp, fp, fp, fp, C.c_uint32, C.c_uint32,
                                            C.POINTER(_Instance), vp, vp, C.c_char_p, sz]
        L.prth_render_model.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(Settings), fp, fp, fp, fp,
                                        C.c_uint32, C.c_uint32, C.c_char_p, C.c_char_p, vp, C.c_char_p, sz]
        L.prth_frame_loop.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(Settings), fp, fp, vp, sz, C.c_uint32, vp,
                                      vp, C.c_char_p, sz]
        L.prth_write_png.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
        i32p, u32p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
        L.prth_checkpoint_write.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
                                            C.c_int32, vp]
        L.prth_checkpoint_read.argtypes = [C.c_char_p, u32p, u32p, u32p, i32p, i32p, i32p, C.POINTER(vp), C.c_char_p, sz]
        _LIB = L
    return _LIB


def default_settings(**kw):
    s = Settings()
    lib().prth_default_settings(C.byref(s))
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def _scene_of(h, name):
    L = lib()
    nv, ni, nm, nt = (getattr(L, "prth_model_num_" + n)(h) for n in ("vertices", "indices", "materials", "textures"))
    v = np.frombuffer(C.string_at(L.prth_model_vertices(h), nv * 48), scenes.VERTEX_DTYPE).copy() if nv else np.zeros(0, scenes.VERTEX_DTYPE)
    i = np.frombuffer(C.string_at(L.prth_model_indices(h), ni * 4), np.uint32).copy() if ni else np.zeros(0, np.uint32)
    m = np.frombuffer(C.string_at(L.prth_model_materials(h), nm * 96), scenes.MATERIAL_DTYPE).copy()
    texs, paths = [], []
    for k in range(nt):
        t, p = _Tex(), C.c_char_p()
        L.prth_model_texture(h, k, C.byref(t), C.byref(p))
        texs.append(np.frombuffer(C.string_at(t.rgba, t.width * t.height * 4), np.uint8).reshape(t.height, t.width, 4).copy())
        paths.append(p.value.decode())
    return {"vertices": v, "indices": i, "materials": m, "textures": texs, "texture_paths": paths,
            "warnings": L.prth_model_warnings(h).decode(errors="replace"), "name": name}


class ModelHandle:
    """A live prth_model (Model of the host layer); close() releases it."""

    def __init__(self, handle, name):
        self.h, self.name = handle, name

    def scene(self):
        return _scene_of(self.h, self.name)

    def close(self):
        if self.h:
            lib().prth_model_free(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def open_model(obj_path):
    h = C.c_void_p()
    err = C.create_string_buffer(1024)
    if lib().prth_model_load(os.fsencode(obj_path), C.byref(h), err, len(err)) != 0:
        raise HostError(err.value.decode(errors="replace"))
    return ModelHandle(h, os.path.basename(obj_path))


def load_model(obj_path):
    """Model(path): returns a scene dict in the layout of pathtracer_amd.scenes (vertices / indices / materials /
    textures as numpy arrays) plus texture paths and loader warnings."""
    with open_model(obj_path) as m:
        return m.scene()


def flatten_instances(models, instances):
    """models: ModelHandles; instances: (modelIndex, 4x4 column-major transform as 16 floats or (4,4) array whose
    [col][row] layout is glm's, instanceId) tuples. Returns (ModelHandle of the flattened scene, firstTriangle)."""
    L = lib()
    hs = (C.c_void_p * len(models))(*[m.h for m in models])
    arr = (_Instance * max(1, len(instances)))()
    for k, (mi, xf, iid) in enumerate(instances):
        arr[k].modelIndex = mi
        arr[k].transform[:] = [float(x) for x in np.asarray(xf, np.float32).reshape(16)]
        arr[k].instanceId = iid
    out = C.c_void_p()
    first = (C.c_uint32 * (len(instances) + 1))()
    err = C.create_string_buffer(1024)
    if L.prth_flatten_instances(hs, len(models), arr, len(instances), C.byref(out), first, err, len(err)) != 0:
        raise HostError(err.value.decode(errors="replace"))
    return ModelHandle(out, "flattened"), np.array(first[:], np.uint32)


def _instance_array(instances):
    arr = (_Instance * max(1, len(instances)))()
    for k, (mi, xf, iid) in enumerate(instances):
        arr[k].modelIndex = mi
        arr[k].transform[:] = [float(x) for x in np.asarray(xf, np.float32).reshape(16)]
        arr[k].instanceId = iid
    return arr


def pool_instances(models, instances):
    """Scene.h poolInstances: (ModelHandle of the object-space pool, BLAS_DTYPE ranges, INSTANCE_DTYPE instances) — the
    arguments of prt_set_scene + prt_set_instances."""
    from .scenes import BLAS_DTYPE, INSTANCE_DTYPE

    hs = (C.c_void_p * len(models))(*[m.h for m in models])
    out = C.c_void_p()
    ranges = np.zeros(len(models), BLAS_DTYPE)
    inst = np.zeros(len(instances), INSTANCE_DTYPE)
    err = C.create_string_buffer(1024)
    if lib().prth_pool_instances(hs, len(models), _instance_array(instances), len(instances), C.byref(out),
                                 ranges.ctypes.data_as(C.c_void_p), inst.ctypes.data_as(C.c_void_p), err, len(err)) != 0:
        raise HostError(err.value.decode(errors="replace"))
    return ModelHandle(out, "pooled"), ranges, inst


def render_instances(models, instances, width, height, settings, eye, target, light=None, first=1, n=1, device=0,
                     flatten=False, moved=None):
    """Renderer::initRaytracing(models, instances, flatten) + n iterations: (accum, primary hit ids of the last one).
    moved: a second instance list whose transforms replace the first after the build (Renderer::updateInstances)."""
    fp = C.POINTER(C.c_float)
    a = lambda x: np.asarray(x, np.float32)
    eye, target = a(eye), a(target)
    lp = ld = None
    if light is not None:
        lp, ld = a(light[0]), a(light[1])
    hs = (C.c_void_p * len(models))(*[m.h for m in models])
    accum = np.zeros((height, width, 4), np.float32)
    ids = np.zeros((height, width), np.uint32)
    err = C.create_string_buffer(1024)
    rc = lib().prth_render_instances(hs, len(models), _instance_array(instances), len(instances), 1 if flatten else 0,
                                     device, width, height, C.byref(settings), eye.ctypes.data_as(fp),
                                     target.ctypes.data_as(fp), lp.ctypes.data_as(fp) if lp is not None else None,
                                     ld.ctypes.data_as(fp) if ld is not None else None, first, n,
                                     _instance_array(moved) if moved is not None else None,
                                     accum.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), err, len(err))
    if rc != 0:
        raise HostError(err.value.decode(errors="replace"))
    return accum, ids


def render_model(model, width, height, settings, eye, target, light=None, first=1, n=1, device=0, resume_from=None,
                 save_to=None):
    fp = C.POINTER(C.c_float)
    a = lambda x: np.asarray(x, np.float32)
    eye, target = a(eye), a(target)
    lp = ld = None
    if light is not None:
        lp, ld = a(light[0]), a(light[1])
    accum = np.zeros((height, width, 4), np.float32)
    err = C.create_string_buffer(1024)
    rc = lib().prth_render_model(model.h, device, width, height, C.byref(settings), eye.ctypes.data_as(fp),
                                 target.ctypes.data_as(fp), lp.ctypes.data_as(fp) if lp is not None else None,
                                 ld.ctypes.data_as(fp) if ld is not None else None, first, n,
                                 os.fsencode(resume_from) if resume_from else None,
                                 os.fsencode(save_to) if save_to else None, accum.ctypes.data_as(C.c_void_p), err, len(err))
    if rc != 0:
        raise HostError(err.value.decode(errors="replace"))
    return accum


class _FrameEvent(C.Structure):
    _fields_ = [("frame", C.c_uint32), ("kind", C.c_int32), ("v", C.c_float * 3)]


class _FrameRecord(C.Structure):
    _fields_ = [("ubo", C.c_uint8 * 464), ("settings_iteration_before", C.c_uint32), ("settings_iteration_after", C.c_uint32),
                ("traced", C.c_int32), ("pad_", C.c_uint32)]


def frame_loop(model, width, height, settings, eye, target, events, n_frames, device=0):
    """prth_frame_loop: n_frames of Renderer::renderFrame with the input `events` [(frame, kind, (x, y, z)), ...].
    Returns (records, accum): records[f] = dict(ubo=464-byte block, before=, after=, traced=)."""
    fp = C.POINTER(C.c_float)
    a = lambda x: np.asarray(x, np.float32)
    eye, target = a(eye), a(target)
    ev = (_FrameEvent * max(1, len(events)))()
    for k, (f, kind, v) in enumerate(events):
        ev[k].frame, ev[k].kind = f, kind
        ev[k].v[:] = [float(x) for x in v]
    rec = (_FrameRecord * n_frames)()
    accum = np.zeros((height, width, 4), np.float32)
    err = C.create_string_buffer(1024)
    rc = lib().prth_frame_loop(model.h, device, width, height, C.byref(settings), eye.ctypes.data_as(fp), target.ctypes.data_as(fp),
                               ev, len(events), n_frames, rec, accum.ctypes.data_as(C.c_void_p), err, len(err))
    if rc != 0:
        raise HostError(err.value.decode(errors="replace"))
    out = []
    for r in rec:
        ubo = np.frombuffer(bytes(r.ubo), dtype=uniforms.UBO_DTYPE)[0].copy()
        out.append(dict(ubo=ubo, before=int(r.settings_iteration_before), after=int(r.settings_iteration_after), traced=bool(r.traced)))
    return out, accum


def write_png(path, rgba):
    rgba = np.ascontiguousarray(rgba, np.uint8)
    h, w = rgba.shape[:2]
    if lib().prth_write_png(os.fsencode(path), rgba.ctypes.data_as(C.c_void_p), w, h) != 0:
        raise HostError(f"cannot write {path}")


def checkpoint_write(path, accum, next_iteration, mode=0, bounces=3, aa_rays=1):
    accum = np.ascontiguousarray(accum, np.float32)
    h, w = accum.shape[:2]
    if lib().prth_checkpoint_write(os.fsencode(path), w, h, next_iteration, mode, bounces, aa_rays,
                                   accum.ctypes.data_as(C.c_void_p)) != 0:
        raise HostError(f"cannot write {path}")


def checkpoint_read(path):
    w, h, it = C.c_uint32(), C.c_uint32(), C.c_uint32()
    mode, bounces, aa = C.c_int32(), C.c_int32(), C.c_int32()
    p = C.c_void_p()
    err = C.create_string_buffer(1024)
    if lib().prth_checkpoint_read(os.fsencode(path), C.byref(w), C.byref(h), C.byref(it), C.byref(mode), C.byref(bounces),
                                  C.byref(aa), C.byref(p), err, len(err)) != 0:
        raise HostError(err.value.decode(errors="replace"))
    try:
        acc = np.frombuffer(C.string_at(p, w.value * h.value * 16), np.float32).reshape(h.value, w.value, 4).copy()
    finally:
        lib().prth_free(p)
    return {"accum": acc, "next_iteration": it.value, "mode": mode.value, "bounces": bounces.value, "aa_rays": aa.value}


def load_image(path):
    w, h, p = C.c_uint32(), C.c_uint32(), C.c_void_p()
    if lib().prth_image_load(os.fsencode(path), C.byref(w), C.byref(h), C.byref(p)) != 0:
        return None
    try:
        return np.frombuffer(C.string_at(p, w.value * h.value * 4), np.uint8).reshape(h.value, w.value, 4).copy()
    finally:
        lib().prth_free(p)


def camera_uniforms(fov, aspect, near, far, pos, yaw, pitch, settings=None):
    s = settings or default_settings()
    ubo = np.zeros((), uniforms.UBO_DTYPE)
    proj = np.zeros(16, np.float32)
    pos = np.asarray(pos, np.float32)
    rc = lib().prth_camera_uniforms(fov, aspect, near, far, pos.ctypes.data_as(C.POINTER(C.c_float)), yaw, pitch, C.byref(s),
                                    ubo.ctypes.data_as(C.c_void_p), proj.ctypes.data_as(C.POINTER(C.c_float)))
    if rc != 0:
        raise HostError("prth_camera_uniforms failed")
    return ubo, proj


def render_obj(obj_path, width, height, settings, eye, target, light=None, first=1, n=1, device=0):
    fp = C.POINTER(C.c_float)
    a = lambda x: np.asarray(x, np.float32)
    eye, target = a(eye), a(target)
    lp = ld = None
    if light is not None:
        lp, ld = a(light[0]), a(light[1])
    accum = np.zeros((height, width, 4), np.float32)
    ubo = np.zeros((), uniforms.UBO_DTYPE)
    err = C.create_string_buffer(1024)
    rc = lib().prth_render_obj(os.fsencode(obj_path), device, width, height, C.byref(settings), eye.ctypes.data_as(fp),
                               target.ctypes.data_as(fp), lp.ctypes.data_as(fp) if lp is not None else None,
                               ld.ctypes.data_as(fp) if ld is not None else None, first, n,
                               accum.ctypes.data_as(C.c_void_p), ubo.ctypes.data_as(C.c_void_p), err, len(err))
    if rc != 0:
        raise HostError(err.value.decode(errors="replace"))
    return accum, ubo
