"""Ad-hoc GPU parity probe (development tool): HIP path vs CPU oracle on small scenes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from pathtracer_amd import scenes, uniforms, capi

def cmp_bvh(ctx, o):
    gn, go = ctx.read_bvh(); on, oo = o.bvh()
    print("  bvh nodes gpu/oracle", len(gn), len(on), "depth", ctx.stats().bvh_depth, o.bvh_depth(), "build ms", ctx.stats().bvh_build_ms)
    if len(gn) == len(on):
        print("  nodes identical:", gn.tobytes() == on.tobytes(), " order identical:", (go == oo).all())
        if gn.tobytes() != on.tobytes():
            bad = [i for i in range(len(gn)) if gn[i].tobytes() != on[i].tobytes()]
            print("  first diffs", bad[:5], gn[bad[0]], on[bad[0]])

def run(name, sc, W, H, mode, iters=2, flags=0, **kw):
    print(f"== {name} {W}x{H} mode {mode} tris {len(sc['indices'])//3}")
    o = po.Oracle(sc, seed=3); o.build_bvh()
    ctx = capi.Context(W, H); ctx.set_scene(sc); ctx.set_seed(3); ctx.build_accel(); ctx.set_flags(flags)
    cmp_bvh(ctx, o)
    acc = np.zeros((H, W, 4), np.float32)
    for it in range(1, iters + 1):
        u = uniforms.make_uniforms(sc, W, H, iteration=it, **kw)
        t = time.time(); acc, ids, rays = o.render(u, W, H, mode, accum=acc, flags=flags, want_ids=True); tc = time.time() - t
        ctx.set_uniforms(u); ctx.reset_stats() ; t = time.time(); ctx.render_iteration(mode); ctx.sync(); tg = time.time() - t
        g = ctx.read_accum(); gid = ctx.read_hit_ids(); st = ctx.stats()
        nd = int((g.view(np.uint32) != acc.view(np.uint32)).any(axis=-1).sum())
        print(f"  it {it}: rays cpu {rays} gpu {st.rays}  ids mismatch {(gid != ids).sum()}  pixels differing {nd}/{W*H}  maxabs {np.nanmax(np.abs(g-acc)):.3g}  cpu {tc:.2f}s gpu {tg*1e3:.1f}ms {st.rays/tg/1e6:.0f} Mrays/s")
    return ctx, o

if __name__ == "__main__":
    sob = capi.Context(8, 8).read_sobol()
    print("sobol table == reference words:", (sob == po.golden_sobol()[:, :32]).all())
    run("cornell", scenes.cornell_box(), 128, 128, 1, ao_rays=16)
    run("cornell", scenes.cornell_box(), 128, 128, 0, bounces=5)
    run("cornell-literal", scenes.cornell_box(), 128, 128, 0, flags=1, bounces=3)
    run("soup", scenes.random_soup(5000), 160, 120, 0, bounces=3)
    run("soup-aa4", scenes.random_soup(2000), 96, 64, 0, bounces=2, aa_rays=4)
    run("soup-rr", scenes.random_soup(2000), 96, 64, 0, flags=2, bounces=8)
    run("conf", scenes.conference_room(detail=0.5), 320, 180, 0, bounces=5)
    run("conf-tex", scenes.conference_room(detail=0.4, textured=True), 320, 180, 0, bounces=5)
