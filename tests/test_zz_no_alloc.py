"""rt_trace_dev_noalloc (the entry point that allocates nothing) against rt_trace_dev, through the C ABI.

Bit-exact on two calls in a row, the second with a smaller input (the static scratch array keeps the first call's
contents: cursors, spill columns, the binning permutation), on scenes that exercise each scratch piece: a deep soup
(capped stack -> spill columns, forced ray binning -> permutation), the BVH2 layout, and the brute-force case (no scratch).
Collected last (file name) so that a failure here cannot hide another test under `pytest -x`.
"""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.float32).reshape(-1).copy()).cuda()


def _hits(torch, n, fill):
    return torch.full((n * 4,), fill, dtype=torch.float32, device="cuda")


@pytest.mark.parametrize("mode,binning,n_tris", [(rt.TRAVERSAL_BVH4Q, 1, 200_000), (rt.TRAVERSAL_BVH4Q, 2, 30_000),
                                                 (rt.TRAVERSAL_BVH2, 1, 30_000), (rt.TRAVERSAL_BRUTE, 0, 300)])
def test_noalloc_equals_trace_dev_two_calls_in_a_row(mode, binning, n_tris):
    import torch
    assert torch.cuda.is_available()
    n1, n2 = (1 << 20) + 4321, 300_017                       # second call smaller: stale scratch must not matter
    if mode == rt.TRAVERSAL_BRUTE:
        n1, n2 = 50_000, 7_001
    tris = scenes.soup(n_tris, half_extent=0.02)
    rays = scenes.random_rays(n1)
    rays["tmax"][::13] = 0.25
    rays["tmax"][::101] = 0.0                                # dead rays
    with rt.Renderer((0,)) as r:
        r.set_option(rt.OPT_TRAVERSAL, mode)
        r.set_option(rt.OPT_RAY_BINNING, binning)
        r.create_data(tris)
        s = torch.cuda.current_stream()
        for n, fill in ((n1, 3.0), (n2, 5.0)):
            d_rays = _dev(torch, rays[:n] if n == n1 else rays[n1 - n:])      # a different slice the second time
            want, got = _hits(torch, n, fill), _hits(torch, n, -fill)
            r.trace_dev(d_rays, n, want, stream=s)
            r.trace_dev_noalloc(d_rays, n, got, stream=s)
            torch.cuda.synchronize()
            assert torch.equal(want.view(torch.int32), got.view(torch.int32)), (mode, binning, n)
            assert (want.view(-1, 4)[:, 3].view(torch.int32) != -1).any()   # some hits


def test_noalloc_rejects_oversize_and_bad_arguments():
    import torch
    tris = scenes.soup(1000, half_extent=0.05)
    with rt.Renderer((0,)) as r:
        with pytest.raises(RuntimeError, match="no scene"):
            r.trace_dev_noalloc(0, 0, 0)
        r.create_data(tris)
        rays = torch.zeros(8, dtype=torch.float32, device="cuda")
        hits = torch.zeros(4, dtype=torch.float32, device="cuda")
        with pytest.raises(RuntimeError, match="RT_NOALLOC_MAX_RAYS"):
            r.trace_dev_noalloc(rays, rt.NOALLOC_MAX_RAYS + 1, hits)         # rejected before anything is launched
        with pytest.raises(RuntimeError, match="null"):
            r.trace_dev_noalloc(None, 1, hits)
        r.trace_dev_noalloc(None, 0, None)                                  # empty buffer: nothing to do
        r.trace_dev_noalloc(rays, 1, hits)
        torch.cuda.synchronize()


def test_noalloc_at_the_maximum_size():
    import torch
    n = rt.NOALLOC_MAX_RAYS
    tris = scenes.soup(50_000, half_extent=0.02)
    rays = scenes.random_rays(n)
    with rt.Renderer((0,)) as r:
        r.set_option(rt.OPT_RAY_BINNING, 1)
        r.create_data(tris)
        d_rays = _dev(torch, rays)
        want, got = _hits(torch, n, 1.0), _hits(torch, n, 2.0)
        r.trace_dev(d_rays, n, want)
        r.trace_dev_noalloc(d_rays, n, got)
        torch.cuda.synchronize()
        assert torch.equal(want.view(torch.int32), got.view(torch.int32))
