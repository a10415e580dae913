"""Multi-GPU path on CPU: world_size-2 gloo run of the sample-sharding + sum-reduce logic that bench.py uses
on RCCL. The renderer in the ranks is the CPU oracle (this is a test of the sharding, not of the kernels)."""
import os
import socket

import numpy as np
import pytest

from pathtracer_amd import scenes, sharding, uniforms


def test_shards_partition_the_iterations():
    for world in (1, 2, 3, 4, 8):
        for total in (0, 1, 5, 8, 1024, 4097):
            got = sorted(i for r in range(world) for i in sharding.shard_iterations(r, world, total))
            assert got == list(range(1, total + 1))
            sizes = [sharding.iteration_shard(r, world, total)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.iteration_shard(2, 2, 4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, W, H, out_dir):
    import torch
    import torch.distributed as dist

    from oracle import pyoracle as po

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.cornell_box()
    o = po.Oracle(sc, seed=5)
    acc = np.zeros((H, W, 4), np.float32)
    rays = 0
    for it in sharding.shard_iterations(rank, world, total):
        u = uniforms.make_uniforms(sc, W, H, bounces=2, iteration=it)
        _, _, r = o.render(u, W, H, 0, accum=acc, threads=2)
        rays += r
    t = torch.from_numpy(acc)
    sharding.sum_reduce_(t, dist)
    rt = torch.tensor([float(rays)], dtype=torch.float64)
    dist.all_reduce(rt)
    if rank == 0:
        np.save(os.path.join(out_dir, "sum.npy"), t.numpy())
        np.save(os.path.join(out_dir, "rays.npy"), rt.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_sum_equals_single_process(po, tmp_path):
    import torch.multiprocessing as mp

    W, H, total, world = 40, 32, 6, 2
    mp.spawn(_worker, args=(world, _free_port(), total, W, H, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sum.npy")
    rays = float(np.load(tmp_path / "rays.npy")[0])
    sc = scenes.cornell_box()
    o = po.Oracle(sc, seed=5)
    ref = np.zeros((H, W, 4), np.float32)
    rr = 0
    for it in range(1, total + 1):
        _, _, r = o.render(uniforms.make_uniforms(sc, W, H, bounces=2, iteration=it), W, H, 0, accum=ref, threads=2)
        rr += r
    assert rays == rr
    assert (got[..., 3] == ref[..., 3]).all()  # vertex counts are integers: exact
    num = ((got - ref)[..., :3] ** 2).mean()
    den = (ref[..., :3] ** 2).mean()
    assert num / den < 1e-10  # fp32 reassociation only


def test_bench_self_launch_starts_the_ranks_and_propagates_failure():
    """`python3 bench.py --gpus 2` with no WORLD_SIZE must start 2 ranks itself (fresh children; the parent never touches
    the GPU). Without a GPU every rank refuses to run (no CPU fallback), and the parent must report that as a failure
    instead of printing a line for fewer GPUs than asked."""
    import subprocess
    import sys

    import torch

    if torch.cuda.is_available():
        pytest.skip("the GPU form of this test is tests/test_gpu_group.py::test_bench_starts_its_own_ranks")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode != 0
    assert "starting 2 ranks" in res.stderr
    assert res.stderr.count("bench.py needs a GPU") >= 2  # both ranks came up and both said so
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]
