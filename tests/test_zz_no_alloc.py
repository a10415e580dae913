"""prt_trace_device_noalloc (include/prt.h): the allocation-free variant of prt_trace_device, whose work-fetch cursor
is a file-scope __device__ array instead of hipMalloc'ed context memory. Through the C-ABI, against prt_trace_device on
the same device-resident rays: hits bit-exact (prim, u, v, t), closest-hit and any-hit, on two calls in a row, the
second with a smaller input (the array keeps the first call's cursor and must be cleared by the entry point)."""
import numpy as np
import pytest

from pathtracer_amd import capi, scenes

pytestmark = pytest.mark.gpu


def _rays(n, seed, extent):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, capi.RAY_DTYPE)
    rays["o"] = rng.uniform(-extent, extent, (n, 3))
    d = rng.uniform(-extent * 0.7, extent * 0.7, (n, 3)) - rays["o"]
    rays["d"] = d * rng.uniform(0.3, 3.0, (n, 1))
    rays["tmin"], rays["tmax"] = 1e-6, 1.0
    return rays


def test_noalloc_trace_equals_trace_device():
    import torch

    sc = scenes.conference_room(detail=0.4)
    ctx = capi.Context(16, 16)
    ctx.set_scene(sc)
    ctx.build_accel()
    dev = "cuda:0"
    warm = False  # the runtime loads its fill kernel lazily at the first hipMemsetAsync: measure free memory after that
    for n, seed in ((200_000, 5), (777, 6)):  # two calls in a row, the second smaller
        host = _rays(n, seed, 5.0)
        rays = torch.from_numpy(host.view(np.float32).reshape(n, 8).copy()).to(dev)
        for any_hit in (False, True):
            ref = torch.full((n, 4), -1.0, dtype=torch.float32, device=dev)
            out = torch.full((n, 4), -2.0, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            ctx.trace_device(rays.data_ptr(), n, ref.data_ptr(), any_hit=any_hit)
            free0 = torch.cuda.mem_get_info()[0]
            ctx.trace_device_noalloc(rays.data_ptr(), n, out.data_ptr(), any_hit=any_hit)
            ctx.sync()
            if warm:
                assert torch.cuda.mem_get_info()[0] == free0, "the entry point changed the amount of free device memory"
            warm = True
            a, b = ref.cpu().numpy().view(np.uint32), out.cpu().numpy().view(np.uint32)
            if any_hit:  # an any-hit query reports *an* occluder: which one depends on the traversal order of the wave
                assert ((a[:, 0] == 0xFFFFFFFF) == (b[:, 0] == 0xFFFFFFFF)).all()
            else:
                assert (a == b).all(), f"n={n}: {(a != b).any(axis=1).sum()} hits differ"
            assert (b[:, 0] != 0xFFFFFFFF).any() and (b[:, 0] == 0xFFFFFFFF).any()
    # on a caller's stream
    n = 5000
    host = _rays(n, 9, 5.0)
    rays = torch.from_numpy(host.view(np.float32).reshape(n, 8).copy()).to(dev)
    ref = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    out = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.trace_device(rays.data_ptr(), n, ref.data_ptr())
    s = torch.cuda.Stream()
    ctx.trace_device_noalloc(rays.data_ptr(), n, out.data_ptr(), stream=s.cuda_stream)
    s.synchronize()
    assert (ref.cpu().numpy().view(np.uint32) == out.cpu().numpy().view(np.uint32)).all()
    # limits and call order
    with pytest.raises(capi.PrtError, match="limit"):
        ctx.trace_device_noalloc(rays.data_ptr(), (1 << 30) + 1, out.data_ptr())
    ctx.trace_device_noalloc(rays.data_ptr(), 0, out.data_ptr())  # empty batch: nothing launched
    ctx.close()
    fresh = capi.Context(8, 8)
    with pytest.raises(capi.PrtError, match="no acceleration structure"):
        fresh.trace_device_noalloc(rays.data_ptr(), n, out.data_ptr())
    fresh.close()
