"""nalu_gpu_dot_static -- the one entry point that allocates nothing (include/nalu_gpu.h) --
against nalu_gpu_dot, through the C ABI: bit-identical, on two calls in a row, the second with a
smaller input (the static scratch keeps the first call's partial sums)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_dot_static_matches_dot():
    import torch
    assert torch.cuda.is_available(), "this test needs the MI355X"
    import nalu_amd
    from nalu_amd import _lib as L
    lib = nalu_amd.load()
    g = nalu_amd.GpuLinearSystem()
    g.beginLinearSystemConstruction(8)
    g.buildNodeGraph(np.arange(8, dtype=np.int32))
    g.finalizeLinearSystem()
    rs = np.random.RandomState(12)
    out = torch.full((2,), -7.0, dtype=torch.float64, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    # 20 M elements: more than 16,384 workgroups x 1,024 elements, so the grid-stride loop and
    # every partial-sum slot are in play; then 3,001 (3 workgroups, the rest of the scratch stale)
    for n in (20_000_000, 3_001, 1, 0):
        a = torch.as_tensor(rs.uniform(-1, 1, n), device="cuda:0")
        b = torch.as_tensor(rs.uniform(-1, 1, n), device="cuda:0")
        want = g.dot(a, b) if n else 0.0  # (nalu_gpu_dot rejects the null pointers of empty tensors)
        L.check(lib.nalu_gpu_dot_static(0, stream, a.data_ptr(), b.data_ptr(), n, out.data_ptr()))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert got[0] == want, (n, got[0], want)  # bit for bit
        assert got[1] == -7.0  # one double written, no more
        ref = float(a.cpu().numpy() @ b.cpu().numpy())
        assert abs(got[0] - ref) <= 1e-12 * max(1.0, float(np.abs(a.cpu().numpy() * b.cpu().numpy()).sum()))
    # sizes beyond the stated maximum and null pointers are rejected before anything is launched
    with pytest.raises(L.NaluGpuError, match="outside"):
        L.check(lib.nalu_gpu_dot_static(0, stream, out.data_ptr(), out.data_ptr(), (1 << 32) + 1, out.data_ptr()))
    with pytest.raises(L.NaluGpuError, match="null"):
        L.check(lib.nalu_gpu_dot_static(0, stream, None, None, 5, out.data_ptr()))
    with pytest.raises(L.NaluGpuError, match="device"):
        L.check(lib.nalu_gpu_dot_static(99, stream, out.data_ptr(), out.data_ptr(), 1, out.data_ptr()))
    g.close()
