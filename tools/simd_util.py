"""SIMD utilisation of the traversal kernels on the bench workload (profiling level 2 counters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import capi, scenes, uniforms
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 5
which = sys.argv[4] if len(sys.argv) > 4 else "conference"  # the bench scenes (bench.py make_scene)
sc = {"conference": lambda: scenes.conference_room(seed=2, detail=1.4), "sponza": lambda: scenes.sponza_scale(),
      "sanmiguel": lambda: scenes.san_miguel_scale()}[which]()
ctx = capi.Context(W, H); ctx.set_scene(sc); ctx.set_seed(2); ctx.build_accel()
ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=B, iteration=1))
ctx.set_profiling(2); ctx.reset_stats(); ctx.render(0, 1, 2, 1)
p = ctx.profile()
for name, rays, nodes, tris, s in (("closest", p.closest_rays, p.closest_nodes, p.closest_tris, p.closest_steps),
                                   ("any", p.any_rays, p.any_nodes, p.any_tris, p.any_steps)):
    wb = rays / 64.0
    print(f"{name}: rays {rays}  nodes/ray {nodes / rays:.2f}  tris/ray {tris / rays:.2f}")
    print(f"   node steps/64 rays {s.node_steps / wb:.1f}  utilisation {s.node_lanes / (64.0 * s.node_steps):.3f}")
    print(f"   tri  steps/64 rays {s.tri_steps / wb:.1f}  utilisation {s.tri_lanes / (64.0 * s.tri_steps):.3f}")
    print(f"   setups/64 rays     {s.setup_steps / wb:.2f}  utilisation {s.setup_lanes / (64.0 * s.setup_steps):.3f}")
    tt = float(s.total_ticks)
    print(f"   wave time: set-up {s.setup_ticks / tt:.3f}  node phases {s.node_ticks / tt:.3f}  triangle phases {s.tri_ticks / tt:.3f}"
          f"   ticks per node step {s.node_ticks / s.node_steps:.0f}  per triangle step {s.tri_ticks / s.tri_steps:.0f}"
          f"  per set-up {s.setup_ticks / s.setup_steps:.0f}")
