#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: Mrays/s (whole job, primary + bounce + shadow rays) on the conference
workload, 1920x1080, path tracing, 5 bounces, GGX + area-light NEE.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no WORLD_SIZE in the environment (no launcher), bench.py starts the N ranks itself (self_launch:
fresh child processes under torch.distributed.run on 127.0.0.1; the parent makes no GPU call) and relays rank 0's line.
LOCAL_RANK maps to device `LOCAL_RANK mod visible devices`, so launchers that isolate one device per rank work too.

A "step" is one progressive iteration (1 sample per pixel over the whole image) per GPU. With N > 1 the
scene is replicated, rank r renders iterations r+1, r+1+N, ... (sample sharding) into its own
float4 buffer and the buffers are summed with one RCCL all-reduce inside the timed region ("weak" scaling:
K iterations per GPU whatever N is). Inputs (scene, BVH, uniforms) are resident in HBM before timing starts.

One JSON line on stdout (rank 0). `roofline` is for the dominant kernel (the fused traversal launch): HBM-side bytes
per launch (PMC FETCH_SIZE / WRITE_SIZE of the committed rocprofv3 passes, per ray, scaled to this run) / the kernel's
own duration (HIP events on the library's stream, one pass in flight, measured in this run after the timed region) /
8 TB/s; `roofline.binding` says what actually bounds the kernel, `roofline.algorithmic` is the SURVEY 8(d) byte
formula on the same duration, `roofline.in_pipeline` the launch duration inside the timed region. `cpu_baseline` is
the CPU oracle (oracle/, a port of the reference shaders — the reference has no CPU path) on all CPUs this process may
use, on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One HIP stream per in-flight iteration + the accumulation stream + torch's own: give every one its own hardware
# queue (the runtime's default is 4 per process; must be set before HIP initialises).
try:
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
except ValueError:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cpus():
    """What this process may use of the host: logical CPUs of the machine, CPUs in its affinity mask, cgroup CPU quota."""
    info = {"cpu_count": os.cpu_count()}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["affinity"] = None
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    info["cgroup_quota"] = quota
    return info


def host_threads():
    """Threads of the CPU baseline: every CPU this process is allowed to run on."""
    h = host_cpus()
    n = h["affinity"] or h["cpu_count"] or 1
    if h["cgroup_quota"]:
        n = min(n, max(1, int(h["cgroup_quota"] + 0.5)))
    return max(1, n)


def load_obj_scene(path, camera=None):
    """A real asset through the host layer's Model (libprt_host: prth_model_load = Model::LoadModelFromFile,
    src/Model.cpp:47-221, with its textures; the reference hard-codes the path, src/vkContext.cpp:52-72). `camera` =
    "ex,ey,ez,tx,ty,tz"; default: inside the bounding box, a quarter of its extent from the centre, looking at the centre.
    As in the reference at start-up the light sits at the camera and faces where it looks (m_moveLight,
    vkContext.cpp:1008-1017)."""
    from pathtracer_amd import hostapi

    scene = hostapi.load_model(path)
    if len(scene["indices"]) == 0:
        raise SystemExit(f"bench.py: {path} holds no triangles")
    p = scene["vertices"]["p"]
    p = p[np.isfinite(p).all(axis=1)]
    lo, hi = p.min(axis=0).astype(np.float64), p.max(axis=0).astype(np.float64)
    if camera:
        c = [float(x) for x in camera.split(",")]
        if len(c) != 6:
            raise SystemExit("--camera wants ex,ey,ez,tx,ty,tz")
        eye, target = np.array(c[:3]), np.array(c[3:])
    else:
        target = (lo + hi) / 2
        eye = target + (hi - lo) * np.array([-0.25, 0.1, 0.22])
    fwd = (target - eye) / max(1e-30, np.linalg.norm(target - eye))
    scene["camera"] = {"eye": eye.astype(np.float32), "target": target.astype(np.float32)}
    scene["light"] = {"pos": eye.astype(np.float32), "dir": fwd.astype(np.float32)}
    scene["name"] = os.path.basename(path)
    return scene


def self_launch(n, argv):
    """`python3 bench.py --gpus N` without a launcher: start N ranks, one process per GPU, exactly as the driver's
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...` form would. This parent has made
    no GPU call (torch is not even imported here): the ranks are fresh child processes, nothing is re-executed.
    Rank 0's JSON line goes to this process's stdout; the exit code is non-zero if any rank failed."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    log(f"[bench] --gpus {n} without WORLD_SIZE: starting {n} ranks on 127.0.0.1:{port}")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:  # relay only rank 0's result line; anything else a rank wrote to stdout goes to stderr
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0 or line is None:
        log(f"[bench] the {n}-rank run failed (exit code {rc}{', no result line' if line is None else ''})")
        return rc or 1
    sys.stdout.write(line)
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=5)
    ap.add_argument("--detail", type=float, default=1.4, help="scene tessellation (1.4 = ~330k triangles)")
    ap.add_argument("--scene", default="conference", choices=["conference", "sponza", "sanmiguel"],
                    help="conference = the metric's workload (default); the others are BASELINE.json configs[3], [4]")
    ap.add_argument("--rr", action="store_true", help="Russian roulette (configs[3])")
    ap.add_argument("--obj", default="", help="render this OBJ file (+ its MTL and textures, loaded by libprt_host's Model) "
                                              "instead of a procedural stand-in: the real conference / Sponza / San Miguel "
                                              "assets when they are supplied (SURVEY.md 8d)")
    ap.add_argument("--camera", default="", help="with --obj: ex,ey,ez,tx,ty,tz (default: inside the bounding box)")
    ap.add_argument("--light-area", type=float, default=0.0, help="half-extent of the area light (default 0.02 = the "
                                                                  "reference's UI default; 0.5 for the large stand-ins)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bounded CPU-baseline sample")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel event timing (roofline = null)")
    ap.add_argument("--pipes", type=int, default=0, help="iterations in flight per GPU (0 = library default)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing "
                                                       "the N > 1 code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: ranks beyond the GPU count share cuda:0")
    ap.add_argument("--dump-accum", default="", help="rank 0 saves the summed accumulation image of the timed region (.npy)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} overrides --gpus {args.gpus}")
    n_gpus = world

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if ndev < local_world and not args.share_gpu:
        # fewer visible devices than ranks on this node: either the launcher gave every rank its own device
        # (HIP_VISIBLE_DEVICES per rank; then ndev == 1 and device 0 is the rank's own), or ranks would share a
        # GPU, which RCCL refuses and which would not be the measurement asked for
        if not (ndev == 1 and any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))):
            raise SystemExit(f"bench.py: {local_world} ranks on this node but {ndev} visible GPU(s); pass --share-gpu "
                             f"(with --backend gloo) only to rehearse the N > 1 code path on a smaller box")
    device_index = local % ndev  # one device per rank; isolated ranks see theirs as device 0
    if rank == 0 and device_index != local:
        log(f"[bench] LOCAL_RANK {local} -> cuda:{device_index} ({ndev} visible device(s))")
    local = device_index
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_

        dist = dist_
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=args.backend)

    from pathtracer_amd import capi, scenes, sharding, uniforms

    W, H, K, WU = args.width, args.height, args.steps, args.warmup
    t0 = time.time()
    if args.obj:
        scene = load_obj_scene(args.obj, args.camera)
        args.scene = scene["name"]
    elif args.scene == "conference":
        scene = scenes.conference_room(seed=2, detail=args.detail)
    elif args.scene == "sponza":
        scene = scenes.sponza_scale(seed=4)
    else:
        scene = scenes.san_miguel_scale(seed=5, million_tris=10.0)
    light_area = args.light_area or (0.02 if (args.scene == "conference" or args.obj) else 0.5)
    ntri = len(scene["indices"]) // 3
    ctx = capi.Context(W, H, device=local)
    ctx.set_scene(scene)
    ctx.set_seed(2)
    ctx.build_accel()
    st = ctx.stats()
    if rank == 0:
        log(f"[bench] scene {scene['name']}: {ntri} triangles, BVH {st.bvh_nodes} nodes depth {st.bvh_depth} "
            f"built on GPU in {st.bvh_build_ms:.1f} ms (setup {time.time() - t0:.1f} s)")
    what = f"OBJ file {scene['name']}" if args.obj else f"{scene['name']} ({args.scene}-scale)"
    workload = (f"{what}, {ntri} triangles, {W}x{H}, path tracing, {args.bounces} bounces"
                f"{' + Russian roulette' if args.rr else ''}, GGX + area-light NEE, 1 spp per step")
    ubo = uniforms.make_uniforms(scene, W, H, bounces=args.bounces, spp=1024, iteration=1, light_area=light_area)
    ctx.set_uniforms(ubo)
    ctx.set_flags(capi.FLAG_RUSSIAN_ROULETTE if args.rr else 0)  # additive accumulation: required for the multi-GPU sum
    if args.pipes:
        ctx.set_pipeline_depth(args.pipes)
    # passes of 24 M pixels or more run one after the other on one stream (prt_render); smaller ones rotate over the pipes
    in_flight = 1 if W * H * min(K, ctx.iteration_batch()) >= (24 << 20) else ctx.pipeline_depth()
    # sample sharding: K iterations per GPU (weak scaling) = iterations 1 .. K*N dealt round-robin
    it_first, it_n, it_stride = sharding.iteration_shard(rank, n_gpus, K * n_gpus)
    assert (it_first, it_n, it_stride) == (1 + rank, K, n_gpus)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local}")
    ctx.bind_accum(accum.data_ptr())

    # instrumented pass (untimed): mean nodes fetched / triangles tested per ray for the bytes formula, and which
    # traversal kernel dominates -- only that kernel is bracketed with events in the timed region (events around
    # every launch cost the pipelined render about a tenth of its rate)
    n_node = n_tri = None
    prof2 = None
    trav_kinds = (capi.K_EXTEND, capi.K_SHADOW, capi.K_SHADOW_EXTEND)
    kind = capi.K_SHADOW_EXTEND
    if not args.no_profile:
        ctx.set_profiling(2)
        ctx.reset_stats()
        ctx.render(capi.MODE_PATH, 1 + rank, max(2, ctx.iteration_batch()), n_gpus)
        prof2 = ctx.profile()
        kind = max(trav_kinds, key=lambda kd: prof2.ms[kd])
        ctx.set_profiling(1)
        ctx.set_profiling_mask(1 << kind)
    # warmup
    ctx.render(capi.MODE_PATH, 1 + rank, WU, n_gpus)
    if WU < ctx.iteration_batch():  # one pass at full size, so that no buffer grows inside the timed region
        ctx.render(capi.MODE_PATH, 1 + rank, ctx.iteration_batch(), n_gpus)
    ctx.sync()
    if dist:
        dist.all_reduce(accum, op=dist.ReduceOp.SUM)  # warm-up of the collective at its real size (untimed)
    accum.zero_()
    torch.cuda.synchronize()
    ctx.reset_stats()

    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    ctx.render(capi.MODE_PATH, it_first, it_n, it_stride)
    ctx.sync()
    sharding.sum_reduce_(accum, dist)  # the one collective of the path: RCCL all-reduce(sum) of W*H*4 floats
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if args.dump_accum and rank == 0:
        np.save(args.dump_accum, accum.cpu().numpy())

    stats = ctx.stats()
    pass_iters = ctx.pass_iterations() or 1  # what the timed render used (the memory policy may have split the passes)
    rays = torch.tensor([float(stats.rays)], dtype=torch.float64, device=f"cuda:{local}")
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
    if dist:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_rays = float(rays.item())
    elapsed = float(tmax.item())
    finite = bool(torch.isfinite(accum).all().item())  # all four channels

    roofline = None
    kernels = {}
    if not args.no_profile:
        p = ctx.profile()
        cn = prof2.closest_nodes / max(1, prof2.closest_rays)
        ct = prof2.closest_tris / max(1, prof2.closest_rays)
        an = prof2.any_nodes / max(1, prof2.any_rays)
        at = prof2.any_tris / max(1, prof2.any_rays)
        # B_ray = 32 + 32 (ray record w+r) + 16 + 16 (hit w+r) + N_node*S_node + N_tri*48,
        # S_node = 64 (BVH4 node with 16-bit boxes, four 16-byte loads)
        b_closest = 96.0 + cn * 64.0 + ct * 48.0
        b_any = 96.0 + an * 64.0 + at * 48.0
        def kind_work(pp, kd):
            """(rays, algorithmic bytes) the launches of kernel kind `kd` processed"""
            fc, fa = pp.fused_closest_rays, pp.fused_any_rays
            if kd == capi.K_EXTEND:
                return pp.closest_rays - fc, (pp.closest_rays - fc) * b_closest
            if kd == capi.K_SHADOW:
                return pp.any_rays - fa, (pp.any_rays - fa) * b_any
            return fc + fa, fc * b_closest + fa * b_any

        names = {capi.K_EXTEND: "k_extend (closest-hit traversal)",
                 capi.K_SHADOW: "k_shadow (any-hit traversal + accumulate)",
                 capi.K_SHADOW_EXTEND: "k_shadow_extend (closest-hit traversal of bounce b+1 and any-hit traversal "
                                       "of bounce b in one launch)"}
        kname = names[kind]
        nrays, nbytes = kind_work(p, kind)
        b_ray = nbytes / max(1, nrays)
        if kind == capi.K_SHADOW_EXTEND:
            fr = p.fused_closest_rays / max(1, nrays)
            nn, nt = fr * cn + (1 - fr) * an, fr * ct + (1 - fr) * at
        else:
            nn, nt = (cn, ct) if kind == capi.K_EXTEND else (an, at)
        launches = max(1, int(p.launches[kind]))
        ms_launch = p.ms[kind] / launches
        rays_launch = nrays / launches
        alg_rate = b_ray * rays_launch / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else 0.0
        in_pipeline = {"ms_per_launch": round(ms_launch, 4), "launches": launches, "rays_per_launch": round(rays_launch, 1),
                       "algorithmic_GB/s": round(alg_rate, 1),
                       "what": "HIP events on the launch's own stream over the timed region; with several passes in flight the "
                               "launch shares the GPU with the other streams' kernels, so this is not the kernel's own duration"}

    # the dominant kernel on its own: same workload with one iteration in flight (untimed extra, after `value`)
    if not args.no_profile:
        ctx.set_pipeline_depth(1)
        ctx.set_profiling_mask(0xffffffff)
        ctx.reset_stats()
        ctx.render(capi.MODE_PATH, 1 + rank, ctx.iteration_batch(), n_gpus)
        p1 = ctx.profile()
        for k, name in capi.KERNEL_NAMES.items():
            if p1.launches[k]:
                kernels[name] = {"ms_total": round(p1.ms[k], 3), "launches": int(p1.launches[k]),
                                 "ms_per_launch": round(p1.ms[k] / p1.launches[k], 4)}
        l1 = max(1, int(p1.launches[kind]))
        ms1 = p1.ms[kind] / l1
        r1, by1 = kind_work(p1, kind)
        n1 = r1 / l1
        alg1 = by1 / l1 / (ms1 * 1e-3) / 1e9 if ms1 > 0 else 0.0
        # HBM-side traffic of the kernel: PMC counters cannot be read inside this process; they come from the committed
        # rocprofv3 passes over this same command (tools/profile.sh -> profiles/pmc_traffic.json), per ray, scaled to
        # this run's rays per launch
        traffic = src = None
        binding = {}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            from pathtracer_amd.buildinfo import kernel_source_hash

            allk = json.load(open(pmc))
            ent = allk.get("scenes", {}).get(args.scene)
            if ent is None or ent.get("workload") != workload:
                src = (f"none: profiles/pmc_traffic.json holds no counters for this workload ({workload!r}); it has "
                       f"{sorted(allk.get('scenes', {}))}")
            elif ent.get("kernel_source_hash") != kernel_source_hash():
                src = (f"none: the counters of scenes[{args.scene!r}] were measured on kernel sources "
                       f"{ent.get('kernel_source_hash')}, this tree is {kernel_source_hash()} (re-run tools/profile.sh + "
                       f"tools/summarize_profile.py)")
            else:
                t = ent["kernels"].get(kname.split(" ")[0], {})
                if t.get("hbm_bytes_per_ray"):
                    traffic = int(t["hbm_bytes_per_ray"] * n1)
                    src = (f"profiles/pmc_traffic.json scenes[{args.scene!r}] [{ent.get('tag', '?')}, kernel sources "
                           f"{ent['kernel_source_hash']}]: (FETCH_SIZE x {t.get('fetch_correction', 2)} + WRITE_SIZE) "
                           f"per ray = {t['hbm_bytes_per_ray']} B at {t.get('rays_per_launch')} rays per launch, scaled to this run's "
                           f"{n1:.0f} rays per launch; counted at the L2's fabric side, so Infinity-Cache hits are included "
                           f"(upper bound on HBM bytes)")
                for key in ("l2_hit_rate", "valu_busy_simd", "valu_active_per_wave_cycle", "wait_any_per_wave_cycle",
                            "tcp_tcc_read_req_per_ray", "tcp_tcc_read_latency_cycles", "l1_miss_lines_per_clk_per_cu",
                            "l1_miss_ceiling_lines_per_clk_per_cu", "verdict"):
                    if key in t:
                        binding[key] = t[key]
        except Exception as e:  # noqa: BLE001
            traffic, src = None, f"none: {e}"
        st2 = prof2.closest_steps if kind == capi.K_EXTEND else prof2.any_steps if kind == capi.K_SHADOW else None
        def lanes(sx):
            return {"node_steps": round(sx.node_lanes / max(1, 64 * sx.node_steps), 3),
                    "triangle_steps": round(sx.tri_lanes / max(1, 64 * sx.tri_steps), 3),
                    "ray_setup": round(sx.setup_lanes / max(1, 64 * sx.setup_steps), 3)}
        binding["lane_utilisation"] = ({"closest": lanes(prof2.closest_steps), "any": lanes(prof2.any_steps)}
                                       if st2 is None else lanes(st2))
        binding["what"] = ("not HBM: the kernel is bound by the latency of dependent, divergent 16-byte gathers (one L2 / "
                           "Infinity-Cache round trip per node visit below the part of the tree that is staged in LDS, the wave "
                           "waits for its slowest lane) at the 7 waves per SIMD its registers and LDS allow; DESIGN.md section 5")
        achieved = traffic / (ms1 * 1e-3) / 1e9 if traffic and ms1 > 0 else None
        roofline = {"bound": "hbm", "kernel": kname,
                    "achieved": None if achieved is None else round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": src,
                    "ms_per_launch": round(ms1, 4), "rays_per_launch": round(n1, 1),
                    "what": "HBM-side bytes of one launch / the kernel's own duration (HIP events, one pass in flight, "
                            "measured in this run) / 8 TB/s",
                    "binding": binding,
                    "algorithmic": {"bytes_per_ray": round(b_ray, 1), "nodes_per_ray": round(nn, 2), "tris_per_ray": round(nt, 2),
                                    "GB/s": round(alg1, 1), "frac_of_hbm_peak": round(alg1 / HBM_PEAK_GBS, 4),
                                    "what": "SURVEY.md 8(d) bytes per ray x rays / the same duration: every node visit counted as "
                                            "64 B although L2 / Infinity Cache serve the re-reads, so this may exceed the HBM peak",
                                    "closest": {"nodes_per_ray": round(cn, 2), "tris_per_ray": round(ct, 2)},
                                    "any": {"nodes_per_ray": round(an, 2), "tris_per_ray": round(at, 2)}},
                    "in_pipeline": in_pipeline}
        accum.zero_()

    cpu_baseline = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        try:
            from oracle import pyoracle as po

            threads = args.cpu_threads or host_threads()
            orc = po.Oracle(scene, seed=2)
            orc.build_bvh()
            t1 = time.perf_counter()
            crays = cit = 0
            while cit < 256 and time.perf_counter() - t1 < args.cpu_seconds:
                cit += 1
                u = uniforms.make_uniforms(scene, W, H, bounces=args.bounces, spp=1024, iteration=cit, light_area=light_area)
                _, _, r = orc.render(u, W, H, 0, flags=capi.FLAG_RUSSIAN_ROULETTE if args.rr else 0, threads=threads)
                crays += r
            dt = time.perf_counter() - t1
            cpu_baseline = {"value": round(crays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                            "sample": f"{cit} iterations (1 spp each, {crays} rays) of the same {W}x{H} {args.scene} "
                                      f"workload, {dt:.1f} s, oracle/liboracle.so on {threads} threads",
                            "host": host_cpus()}
        except Exception as e:  # the baseline is a report, never a dependency of the GPU path
            cpu_baseline = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}

    if rank == 0:
        value = total_rays / elapsed / 1e6
        out = {
            "metric": f"Mrays/s (whole node, incl. secondary) {args.scene} {'1080p' if (W, H) == (1920, 1080) else f'{W}x{H}'}",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": n_gpus,
            "steps": K,
            "warmup": WU,
            "ms_per_step": round(elapsed / K * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": (f"OBJ asset {args.obj} loaded through libprt_host (Model)" if args.obj else
                     f"synthetic (procedural {args.scene}-scale scene; the reference's scenes are not in its repo)"),
            "config": {"workload": workload,
                       "triangles": ntri, "width": W, "height": H, "bounces": args.bounces,
                       "iterations_per_gpu": K, "parallelism": f"spp-sharded x{n_gpus}" if n_gpus > 1 else "single GPU",
                       "rays_total": int(total_rays), "rays_per_sample": round(total_rays / (K * n_gpus * W * H), 3),
                       "bvh_build_ms": round(st.bvh_build_ms, 2), "image_finite": finite,
                       "iterations_in_flight": in_flight, "iterations_per_pass": pass_iters, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "kernels_standalone": kernels,
        }
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
