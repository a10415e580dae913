"""prt_group_* (include/prt.h): the single-process, several-contexts form of the sample sharding (SURVEY.md 8b
"Multi-GPU", 8e). On the one-GPU test box both members sit on device 0, which exercises the sharding arithmetic,
the host threads, the copy-and-add exchange and the clear-after-reduce rule; the RCCL exchange (distinct devices)
is exercised with a one-member communicator. Bar: the reduced image equals, bit for bit, the fp32 sum of the
members' images (root + others in member order); against an unsharded render only the summation order differs."""
import os

import numpy as np
import pytest

from pathtracer_amd import capi, scenes, uniforms

pytestmark = pytest.mark.gpu


def _single(sc, W, H, u, iters, seed=5):
    c = capi.Context(W, H)
    c.set_scene(sc)
    c.set_seed(seed)
    c.build_accel()
    c.set_uniforms(u)
    for first, n, stride in iters:
        c.render(capi.MODE_PATH, first, n, stride)
    out = c.read_accum()
    rays = c.stats().rays
    c.close()
    return out, rays


@pytest.mark.parametrize("members", [2, 3])
def test_group_on_one_device_equals_member_sums(members):
    sc = scenes.cornell_box()
    W, H = 80, 60
    u = uniforms.make_uniforms(sc, W, H, iteration=1, bounces=3)
    g = capi.Group([0] * members, W, H)
    assert g.size() == members and not g.uses_rccl()
    g.set_scene(sc)
    g.set_seed(5)
    g.set_uniforms(u)
    g.render(capi.MODE_PATH, 1, 7)  # 7 iterations over 2 or 3 members: uneven shares
    g.reduce(0)
    got = g.read_accum(0)
    parts, rays = [], 0
    for r in range(members):
        cnt = (7 - r + members - 1) // members
        p, n = _single(sc, W, H, u, [(1 + r, cnt, members)])
        parts.append(p)
        rays += n
    exp = parts[0].copy()
    for p in parts[1:]:
        exp = exp + p
    assert got.tobytes() == exp.tobytes()
    assert g.stats().rays == rays and g.stats().iterations == 7
    whole, _ = _single(sc, W, H, u, [(1, 7, 1)])
    assert np.allclose(got, whole, rtol=2e-6, atol=1e-6)  # fp32 summation order only
    # second round: the root keeps the total, the other members were cleared
    g.render(capi.MODE_PATH, 8, 4)
    g.reduce(0)
    got2 = g.read_accum(0)
    whole2, _ = _single(sc, W, H, u, [(1, 11, 1)])
    assert np.allclose(got2, whole2, rtol=3e-6, atol=1e-6)
    assert (got2[..., 3] == whole2[..., 3]).all()  # the sample counts are small integers: exact in any order
    g.close()


def test_group_with_instances_equals_single_context():
    """prt_group_set_instances: a two-level scene on every member; the sharded sum differs from one context's render by
    the fp32 summation order only."""
    sc = scenes.instanced_scene(seed=23, n_instances=12, wild=True)
    W, H = 64, 48
    u = uniforms.make_uniforms(sc, W, H, iteration=1, bounces=3)
    g = capi.Group([0, 0], W, H)
    g.set_scene(sc)
    g.set_instances(sc["blas"], sc["instances"])
    g.set_seed(5)
    g.set_uniforms(u)
    g.render(capi.MODE_PATH, 1, 6)
    g.reduce(0)
    got = g.read_accum(0)
    c = capi.Context(W, H)
    c.set_scene(sc)
    c.set_instances(sc["blas"], sc["instances"])
    c.set_seed(5)
    c.build_accel()
    c.set_uniforms(u)
    c.render(capi.MODE_PATH, 1, 6)
    whole = c.read_accum()
    assert g.stats().rays == c.stats().rays
    assert g.stats().trace_overruns == 0 and c.stats().trace_overruns == 0
    c.close()
    g.close()
    assert np.allclose(got, whole, rtol=3e-6, atol=1e-6) and (got[..., 3] == whole[..., 3]).all()


def test_group_reduce_to_other_root_and_errors():
    sc = scenes.cornell_box()
    W, H = 48, 32
    u = uniforms.make_uniforms(sc, W, H, iteration=1, bounces=2)
    g = capi.Group([0, 0], W, H)
    with pytest.raises(capi.PrtError):
        g.render(capi.MODE_PATH, 1, 2)  # no scene yet: the member's message comes through
    g.set_scene(sc)
    g.set_seed(5)
    g.set_uniforms(u)
    g.render(capi.MODE_PATH, 1, 4)
    g.reduce(1)
    a = g.read_accum(1)
    assert a[..., 3].max() >= 4.0 and (g.read_accum(0) == 0).all()
    with pytest.raises(capi.PrtError):
        g.reduce(2)
    g.close()
    with pytest.raises(capi.PrtError):
        capi.Group([], W, H)
    with pytest.raises(capi.PrtError):
        capi.Group([999], W, H)


def test_group_rccl_plumbing_single_member():
    """One member, RCCL forced: librccl.so is found, a communicator is made, ncclReduce runs in place."""
    os.environ["PRT_GROUP_FORCE_RCCL"] = "1"
    try:
        sc = scenes.cornell_box()
        W, H = 48, 32
        u = uniforms.make_uniforms(sc, W, H, iteration=1, bounces=2)
        g = capi.Group([0], W, H)
        assert g.uses_rccl()
        g.set_scene(sc)
        g.set_seed(5)
        g.set_uniforms(u)
        g.render(capi.MODE_PATH, 1, 3)
        g.reduce(0)
        got = g.read_accum(0)
        exp, _ = _single(sc, W, H, u, [(1, 3, 1)])
        assert got.tobytes() == exp.tobytes()
        g.close()
    finally:
        del os.environ["PRT_GROUP_FORCE_RCCL"]


def _bench_two_ranks(tmp_path, launcher, W, H, K, detail, extra=()):
    import json
    import socket
    import subprocess
    import sys

    from pathtracer_amd import uniforms

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "sum.npy"
    bench_args = ["--gpus", "2", "--share-gpu", "--backend", "gloo", "--steps", str(K), "--warmup", "1", "--no-cpu-baseline",
                  *extra, "--dump-accum", str(out)]
    if launcher == "torchrun":
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py")] + bench_args
    else:  # the form in the driver's BENCH_rNN.cmd: no launcher, bench.py starts its ranks itself
        cmd = ["python3", "bench.py"] + bench_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]  # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["parallelism"] == "spp-sharded x2"
    assert line["steps"] == K and line["config"]["iterations_per_gpu"] == K
    got = np.load(out)
    sc = scenes.conference_room(seed=2, detail=detail)
    with capi.Context(W, H) as ctx:
        ctx.set_scene(sc)
        ctx.set_seed(2)
        ctx.build_accel()
        ctx.set_uniforms(uniforms.make_uniforms(sc, W, H, bounces=5, spp=1024, iteration=1, light_area=0.02))
        ctx.render(capi.MODE_PATH, 1, 2 * K)
        ref = ctx.read_accum()
        assert line["config"]["rays_total"] == ctx.stats().rays
    assert (got[..., 3] == ref[..., 3]).all()  # vertex counts: exact
    num = ((got - ref)[..., :3].astype(np.float64) ** 2).mean()
    den = (ref[..., :3].astype(np.float64) ** 2).mean()
    assert num / den < 1e-10
    return line


def test_bench_two_ranks_on_one_gpu_equal_a_single_process(tmp_path):
    """The N > 1 code path of bench.py — one process per rank, pathtracer_amd.sharding's round-robin deal, ONE all-reduce —
    run under torch.distributed.run, with two ranks sharing GPU 0 and the gloo backend because the box has one GPU. The
    summed image equals a single-process render of the same iterations up to fp32 summation order.
    What this cannot cover: the RCCL transport between two distinct devices."""
    _bench_two_ranks(tmp_path, "torchrun", 192, 108, 5, 0.3,
                     ("--no-profile", "--width", "192", "--height", "108", "--detail", "0.3"))


def test_bench_starts_its_own_ranks(tmp_path):
    """`python3 bench.py --gpus 2 ...` with no launcher and no WORLD_SIZE (the command form of the driver's BENCH_rNN.cmd)
    starts two ranks itself — at the metric's own size: 328 k triangles, 1920x1080, roofline leg included — and prints
    one line with n_gpus == 2 whose ray count and summed image are those of a single process."""
    line = _bench_two_ranks(tmp_path, "self", 1920, 1080, 4, 1.4)
    assert line["roofline"] and line["roofline"]["ms_per_launch"] > 0
