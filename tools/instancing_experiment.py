"""Two-level traversal (prt_set_instances) against the same scene flattened into one BVH: rate, build time, node count.
usage: python tools/instancing_experiment.py [n_instances] [detail] [width height]   (measurement tool, GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
import numpy as np
from pathtracer_amd import capi, scenes, uniforms
import instancing

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
detail = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
sc = scenes.instanced_scene(seed=31, n_instances=n_inst, detail=detail, wild=False)
flat = instancing.flatten(sc)
print(f"pool {len(sc['indices']) // 3} triangles in {len(sc['blas'])} structures, {len(sc['instances'])} instances = "
      f"{len(flat['indices']) // 3} instanced triangles; {W}x{H}, 5 bounces", flush=True)
for name, scene, two in (("two-level", sc, True), ("flattened", flat, False)):
    ctx = capi.Context(W, H)
    try:
        ctx.set_scene(scene)
    except capi.PrtError as e:
        print(f"{name:10s} refused: {e}", flush=True)
        ctx.close()
        continue
    if two:
        ctx.set_instances(sc["blas"], sc["instances"])
    ctx.build_accel()
    st = ctx.stats()
    ctx.set_uniforms(uniforms.make_uniforms(scene, W, H, bounces=5, iteration=1, light_area=0.5))
    ctx.render(capi.MODE_PATH, 1, 8)
    ctx.sync(); ctx.reset_stats()
    t0 = time.time()
    ctx.render(capi.MODE_PATH, 9, 32)
    ctx.sync()
    dt = time.time() - t0
    rays = ctx.stats().rays
    ctx.set_profiling(2); ctx.reset_stats(); ctx.render(capi.MODE_PATH, 1, 2); p = ctx.profile()
    nr = p.closest_rays + p.any_rays
    print(f"{name:10s} build {st.bvh_build_ms:8.1f} ms  {st.bvh_inner_nodes:9d} BVH4 nodes ({st.bvh_inner_nodes * 64 / 1e6:7.1f} MB)  "
          f"{rays / dt / 1e6:8.1f} Mrays/s  {(p.closest_nodes + p.any_nodes) / nr:5.1f} nodes + {(p.closest_tris + p.any_tris) / nr:4.1f} tris per ray",
          flush=True)
    ctx.set_profiling(1); ctx.set_pipeline_depth(1); ctx.reset_stats(); ctx.render(capi.MODE_PATH, 1, 32); p = ctx.profile()
    print("           kernel ms per pass of 32 iterations: " + ", ".join(
        f"{capi.KERNEL_NAMES[k]} {p.ms[k]:.2f}" for k in capi.KERNEL_NAMES if p.launches[k]), flush=True)
    ctx.set_profiling(0)
    if two:
        t0 = time.time()
        ctx.update_instance_transforms(sc["instances"]["transform"])
        print(f"           instance transform update (top level rebuilt): {(time.time() - t0) * 1e3:.1f} ms", flush=True)
    ctx.close()
