import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pathtracer_amd import capi, scenes, uniforms
free, total = torch.cuda.mem_get_info()
print("free GB", free / 2**30, "total", total / 2**30, flush=True)
W, H = 1920, 1080
sc = scenes.conference_room(seed=2, detail=0.5)
ref = None
for hog_gb in (0, None):
    hog = None
    if hog_gb is None:
        free, _ = torch.cuda.mem_get_info()
        keep = 14 * 2**30  # leave room for a pass of ~16 iterations (0.6 GB each) + scene
        hog = torch.empty(int(free - keep), dtype=torch.uint8, device="cuda")
        print("hog GB", hog.numel() / 2**30, "free now", torch.cuda.mem_get_info()[0] / 2**30, flush=True)
    with capi.Context(W, H) as ctx:
        ctx.set_scene(sc); ctx.set_seed(3); ctx.build_accel()
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=5))
        ctx.render(capi.MODE_PATH, 1, 64, 1)
        ctx.sync()
        img = ctx.read_accum()
        print("pass iterations used:", ctx.pass_iterations(), "rays", ctx.stats().rays, flush=True)
        if ref is None: ref = (img.tobytes(), ctx.stats().rays)
        else: print("same image and rays as the unconstrained render:", (img.tobytes(), ctx.stats().rays) == ref, flush=True)
    del hog
