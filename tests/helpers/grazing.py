"""Test inputs for the grazing-ray case of the watertight test (shared by the CPU and the GPU tests)."""
import numpy as np

from pathtracer_amd import scenes


def grazing_rays(ray_dtype, sc, n, seed):
    """Bounce-style rays (|d| = 1000, tmin 1e-6) that start ON a triangle and leave it almost in its plane — the
    configuration in which the edge functions of the watertight test are rounding noise."""

    rng = np.random.default_rng(seed)
    P = sc["vertices"]["p"][sc["indices"]].reshape(-1, 3, 3).astype(np.float64)
    t = rng.integers(0, len(P), n)
    w = rng.dirichlet((1, 1, 1), n)
    # half of the rays start within ~1e-6 .. 1e-3 (relative) of an edge, so that the ray can have left the triangle's
    # box before t = tmin while the noisy edge functions still call it a hit
    near = rng.random(n) < 0.5
    k = rng.integers(0, 3, n)
    w[near, k[near]] *= 10.0 ** rng.uniform(-6, -3, near.sum())
    w /= w.sum(axis=1, keepdims=True)
    o3 = (P[t] * w[:, :, None]).sum(axis=1)
    e1, e2 = P[t, 1] - P[t, 0], P[t, 2] - P[t, 0]
    nrm = np.cross(e1, e2)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    inpl = e1 * rng.normal(size=(n, 1)) + e2 * rng.normal(size=(n, 1))
    inpl /= np.maximum(np.linalg.norm(inpl, axis=1, keepdims=True), 1e-30)
    d3 = inpl + nrm * (10.0 ** rng.uniform(-6, -2, (n, 1))) * rng.choice([-1.0, 1.0], (n, 1))
    rays = np.zeros(n, ray_dtype)
    rays["o"], rays["d"] = o3.astype(np.float32), (d3 / np.linalg.norm(d3, axis=1, keepdims=True) * 1000.0).astype(np.float32)
    rays["tmin"], rays["tmax"] = 1e-6, 1.0
    return rays




def strip_scene():
    """Thin slanted strips (like the ribs of a column) in a triangle soup."""
    mb = scenes.MeshBuilder()
    rng = np.random.default_rng(3)
    for k in range(40):  # thin slanted strips, like the ribs of a column
        x, z = rng.uniform(-6, 6), rng.uniform(-6, 6)
        mb.add(np.array([[[x, 4.4, z + 0.02], [x, 9.4, z + 0.02], [x - 0.02, 9.4, z]],
                         [[x, 4.4, z + 0.02], [x - 0.02, 9.4, z], [x - 0.02, 4.4, z]]], np.float32), 0)
    soup = scenes.random_soup(1500, seed=9)
    mb.add(soup["vertices"]["p"][soup["indices"]].reshape(-1, 3, 3), 0)
    sc = mb.finish([scenes.make_material()])
    return sc
