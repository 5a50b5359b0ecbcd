"""The entry point that allocates nothing (mtsamd_trace_device_static) against mtsamd_trace_device, through the C ABI:
two calls in a row, the second with a smaller input (the static array still holds the first call's hits), bit-equal hit
records; the size limit is enforced."""
import numpy as np
import pytest

from conftest import random_rays
from mitsuba_amd import scene as S

pytestmark = pytest.mark.gpu


def test_static_scratch_query_matches_the_allocating_one(gpu):
    import torch
    sd = S.small_test_scene()
    g = gpu.Scene(sd)
    limit = gpu.trace_device_static_max()
    assert limit == 1 << 20
    for n, seed in ((300000, 21), (70001, 22), (1, 23)):
        rays = random_rays(sd, n, seed=seed)
        d_rays = torch.tensor(rays, device="cuda")
        a = torch.full((n, 5), -1.0, device="cuda", dtype=torch.float32)
        b = torch.full((n, 5), -2.0, device="cuda", dtype=torch.float32)
        g.trace_device(d_rays.data_ptr(), n, a.data_ptr())
        g.trace_device_static(d_rays.data_ptr(), n, b.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), n
        assert (a[:, 0] < 1e30).float().mean() > 0.05 or n == 1
    g.trace_device_static(0, 0, 0)              # nothing to do, nothing touched
    big = torch.zeros(8, device="cuda")
    with pytest.raises(gpu.MtsamdError, match="at most"):
        g.trace_device_static(big.data_ptr(), limit + 1, big.data_ptr())
    # a second scene on the same device shares the array: sequential calls stay correct
    g2 = gpu.Scene(S.cornell_box(32, 32))
    rays = random_rays(g2.sd, 5000, seed=5)
    d_rays = torch.tensor(rays, device="cuda")
    a = torch.zeros(5000, 5, device="cuda")
    b = torch.zeros(5000, 5, device="cuda")
    g2.trace_device(d_rays.data_ptr(), 5000, a.data_ptr())
    g2.trace_device_static(d_rays.data_ptr(), 5000, b.data_ptr())
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
