"""Sample sharding for multi-GPU runs: the scene is replicated, iterations are dealt
round-robin, per-rank float4 buffers are summed once. The reference has no multi-GPU path."""
from __future__ import annotations


def iteration_shard(rank: int, world: int, total_iterations: int, first: int = 1):
    """Iterations first .. first+total-1 dealt round-robin: rank r gets first+r, first+r+world, ...
    Returns (first_iter, n_iter, stride) for prt_render()."""
    if world < 1 or not (0 <= rank < world) or total_iterations < 0:
        raise ValueError("bad shard arguments")
    n = (total_iterations - rank + world - 1) // world if total_iterations > rank else 0
    return first + rank, n, world


def shard_iterations(rank: int, world: int, total_iterations: int, first: int = 1):
    f, n, s = iteration_shard(rank, world, total_iterations, first)
    return [f + k * s for k in range(n)]


def sum_reduce_(tensor, dist=None):
    """The one collective of the path: sum the per-rank accumulation buffers in place (RCCL on GPUs)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor
