"""Python mirror of the reference's parameter block and of the host code that fills it.

* ``UBO_DTYPE``            <- vkContext::UniformBufferObject (/root/reference/src/vkContext.h:54-82)
* ``perspective/look_at``  <- glm::perspective (RH, depth 0..1) and glm::lookAt as CameraControls
                              uses them (/root/reference/src/CameraControls.cpp:35-41, 111-126)
* ``make_uniforms``        <- vkContext::updateGraphicsUniforms (/root/reference/src/vkContext.cpp:993-1060)

The C++ host layer (pathtracer_amd/host) is the drop-in implementation; this module exists so
that tests and bench.py can prepare the same 464 bytes without a compiler in the loop.
"""
from __future__ import annotations

import numpy as np

UBO_DTYPE = np.dtype(
    [
        ("model", "<f4", 16),
        ("view", "<f4", 16),
        ("proj", "<f4", 16),
        ("modelIT", "<f4", 16),
        ("projViewInverse", "<f4", 16),
        ("lightTransform", "<f4", 16),
        ("lightSize", "<f4", 2),
        ("pad0", "<f4", 2),
        ("lightE", "<f4", 3),
        ("pad1", "<f4"),
        ("numIndirectBounces", "<i4"),
        ("samplesPerPixel", "<i4"),
        ("lightSourceArea", "<f4"),
        ("lightOtherE", "<f4"),
        ("numAArays", "<i4"),
        ("filterRadius", "<f4"),
        ("numAOrays", "<i4"),
        ("aoRayLength", "<f4"),
        ("iteration", "<u4"),
        ("time", "<f4"),
        ("pad2", "<f4", 2),
    ]
)
assert UBO_DTYPE.itemsize == 464


def _norm(v):
    return v / np.linalg.norm(v)


def perspective(fovy_rad, aspect, near, far):
    """glm::perspectiveRH_ZO; returned as a 4x4 with M[row, col]."""
    t = np.tan(fovy_rad / 2.0)
    m = np.zeros((4, 4))
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = 1.0 / t
    m[2, 2] = far / (near - far)
    m[3, 2] = -1.0
    m[2, 3] = -(far * near) / (far - near)
    return m


def look_at(eye, center, up=(0.0, 1.0, 0.0)):
    """glm::lookAtRH."""
    eye, center, up = (np.asarray(a, np.float64) for a in (eye, center, up))
    f = _norm(center - eye)
    s = _norm(np.cross(f, up))
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -s @ eye, -u @ eye, f @ eye
    return m


def _colmajor(m):
    return np.asarray(m, np.float64).T.reshape(16).astype(np.float32)


def light_transform(pos, direction):
    """A camera-to-world style matrix whose -Z axis is `direction` (the reference captures
    inverse(view), so the light faces along -xform[2]; vkContext.cpp:1013-1017, AreaLight.h:26)."""
    d = _norm(np.asarray(direction, np.float64))
    up = np.array([0.0, 1.0, 0.0]) if abs(d[1]) < 0.99 else np.array([0.0, 0.0, -1.0])
    z = -d
    x = _norm(np.cross(up, z))
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, np.asarray(pos, np.float64)
    return m


def make_uniforms(scene, width, height, *, bounces=3, spp=128, aa_rays=1, filter_radius=1.0, ao_rays=16,
                  ao_length=1.0, light_area=0.02, light_e=100.0, light_other_e=1.0, iteration=1,
                  fov_deg=65.0, near=0.01, far=100.0):
    """Fill the 464-byte block the way updateGraphicsUniforms does (defaults = vkContext.h:161-192)."""
    u = np.zeros((), UBO_DTYPE)
    cam = scene["camera"]
    view = look_at(cam["eye"], cam["target"])
    proj = perspective(np.radians(fov_deg), width / height, near, far)
    proj[1, 1] *= -1.0  # vkContext.cpp:1003
    u["model"] = _colmajor(np.eye(4))
    u["view"] = _colmajor(view)
    u["proj"] = _colmajor(proj)
    u["modelIT"] = _colmajor(np.eye(4))
    u["projViewInverse"] = _colmajor(np.linalg.inv(view) @ np.linalg.inv(proj))  # vkContext.cpp:1006
    u["lightTransform"] = _colmajor(light_transform(scene["light"]["pos"], scene["light"]["dir"]))
    u["lightSize"] = (light_area, light_area)
    u["lightE"] = (light_e, light_e, light_e)
    u["numIndirectBounces"] = bounces
    u["samplesPerPixel"] = spp
    u["lightSourceArea"] = light_area
    u["lightOtherE"] = light_other_e
    u["numAArays"] = aa_rays
    u["filterRadius"] = filter_radius
    u["numAOrays"] = ao_rays
    u["aoRayLength"] = ao_length
    u["iteration"] = iteration
    u["time"] = 0.0
    return u
