"""Free device memory after repeated create-render-destroy rounds (does prt_destroy give everything back?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import capi, scenes, uniforms
sc = scenes.cornell_box()
W, H = 256, 192
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
prev = None
for k in range(30):
    ctx = capi.Context(W, H)
    if mode != "create":
        ctx.set_scene(sc); ctx.build_accel()
    if mode == "full":
        ctx.set_pipeline_depth(1 + k % 3)
        ctx.set_iteration_batch(1 + (3 * k) % 8)
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=2 + k % 3))
        ctx.render(capi.MODE_PATH, 1, 9, 1)
        ctx.read_accum()
    ctx.close()
    torch.cuda.synchronize()
    free = torch.cuda.mem_get_info()[0]
    print(k, free >> 20, "MiB free", "" if prev is None else f"delta {(prev - free) >> 10} KiB", flush=True)
    prev = free
