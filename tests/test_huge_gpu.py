"""GPU: tensors of MORE than 2^31 elements -- every offset in the kernels is 64-bit (row index x row length, tile index x tile size,
the ragged tail's position); nothing in the other GPU tests is big enough to notice a 32-bit product.  288 GB of HBM make the size
cheap: 4.3 GB per bf16 tensor.  No oracle run at this size (minutes of CPU): the properties are size-independent --
  * the op on the whole tensor == the op on its two halves (each below 2^31 elements), y and dx bit for bit;
  * the un-rounded fp64 sums of the halves, computed with the WHOLE tensor's element count in the gradient scaler, add up to the
    whole tensor's (the multi-GPU contract, here across the 2^31 boundary);
  * the first and the last million elements (the ragged tail included) against the oracle, bit for bit.
"""
import numpy as np
import pytest
import torch

from oracle import lsq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torchlsq  # noqa: F401
    from torchlsq import extension
    extension._assert_has_ops()
    assert torch.cuda.is_available()
    if torch.cuda.get_device_properties(0).total_memory < (60 << 30):
        pytest.skip("needs ~30 GB of device memory")
    return torch.device("cuda:0")


def _fill(n, seed, mean, std, dev):
    from torchlsq import synth
    return synth.normal_like(n, seed, mean, std, device=dev, dtype=torch.bfloat16, chunk=1 << 26)


def test_per_tensor_beyond_2_31_elements(dev):
    n = (1 << 31) + 4099                       # odd: a ragged tail behind the last packet, past the 32-bit boundary
    x, g = _fill(n, 71, 1.5, 1.0, dev), _fill(n, 72, 0.0, 1e-3, dev)
    s, b = torch.tensor([0.03], device=dev), torch.tensor([0.0], device=dev)
    q = (0, 127, 0, 255, True, 1.0, False, False, False)
    ops = torch.ops.torchlsq_native
    y = ops.lsq_forward_per_tensor(x, s, b, *q)
    dx, wide = ops.lsq_backward_per_tensor_wide(g, x, s, b, *q, n)
    h = (1 << 30) + 24                         # the halves: 2^30 + 24 and 2^30 + 4075 elements, both 16-byte aligned starts
    for lo, hi in ((0, h), (h, n)):
        ya = ops.lsq_forward_per_tensor(x[lo:hi], s, b, *q)
        assert torch.equal(ya, y[lo:hi]), "y differs in [%d, %d)" % (lo, hi)
        del ya
    parts = []
    for lo, hi in ((0, h), (h, n)):
        da, wa = ops.lsq_backward_per_tensor_wide(g[lo:hi], x[lo:hi], s, b, *q, n)
        assert torch.equal(da, dx[lo:hi]), "dx differs in [%d, %d)" % (lo, hi)
        parts.append(wa)
        del da
    np.testing.assert_allclose((parts[0] + parts[1]).cpu().numpy(), wide.cpu().numpy(), rtol=1e-12, atol=0)
    # the two ends against the oracle (fp32 math on the stored values): the head, and the tail with its ragged last elements
    m = 1 << 20
    for lo, hi in ((0, m), (n - m - 3, n)):
        xn, gn = x[lo:hi].float().cpu().numpy(), g[lo:hi].float().cpu().numpy()
        oy = O.fwd_pt(xn, 0.03, 0.0, 0, 127, 0, 255)
        r = O.bwd_pt(gn, xn, 0.03, 0.0, 0, 127, 0, 255, True, 1.0, False, numel_for_scaler=n)
        assert torch.equal(y[lo:hi].cpu(), torch.from_numpy(oy).to(torch.bfloat16)), "y vs oracle at [%d, %d)" % (lo, hi)
        assert torch.equal(dx[lo:hi].cpu(), torch.from_numpy(r.dx).to(torch.bfloat16)), "dx vs oracle at [%d, %d)" % (lo, hi)


def test_per_channel_beyond_2_31_elements(dev):
    shape = (41, 1024, 228, 228)               # 2.18e9 elements: NCHW, 256-lane windows; row length 53 M, 41 rows
    n = int(np.prod(shape))
    assert n > (1 << 31)
    x, g = _fill(n, 73, 0.2, 1.0, dev).view(shape), _fill(n, 74, 0.0, 1e-3, dev).view(shape)
    from torchlsq import synth
    s, b = synth.uniform_like(1024, 75, 0.02, 0.2, device=dev), synth.normal_like(1024, 76, 0.0, 0.1, device=dev)
    q = (-8, 7, -128, 127, True, 1.0, False, False, False)
    ops = torch.ops.torchlsq_native
    y = ops.lsq_forward_per_channel(x, s, b, 1, *q)
    dx, wide = ops.lsq_backward_per_channel_wide(g, x, s, b, 1, *q, n)
    parts = []
    for lo, hi in ((0, 20), (20, 41)):
        assert torch.equal(ops.lsq_forward_per_channel(x[lo:hi], s, b, 1, *q), y[lo:hi]), "y differs in rows [%d, %d)" % (lo, hi)
        da, wa = ops.lsq_backward_per_channel_wide(g[lo:hi], x[lo:hi], s, b, 1, *q, n)
        assert torch.equal(da, dx[lo:hi]), "dx differs in rows [%d, %d)" % (lo, hi)
        parts.append(wa)
        del da
    tot = (parts[0] + parts[1]).cpu().numpy()
    # (per channel the halves' sums are added in another order than the whole's partial rows: fp64 rounding, not more;
    #  16-bit storage adds up to four rows in fp32 first -- include/lsq_hip.h -- and the row groups of 4 differ between the
    #  whole and the halves: 1e-6 of the sum here, the bar itself is held against the oracle elsewhere)
    np.testing.assert_allclose(tot, wide.cpu().numpy(), rtol=1e-6, atol=1e-12)
    # the last image against the oracle, bit for bit (its offset is beyond 2^31 elements)
    xn, gn = x[40:41].float().cpu().numpy(), g[40:41].float().cpu().numpy()
    outer, C, inner = O.axis_to_ocl(xn.shape, 1)
    oy = O.fwd_pc(xn, s.cpu().numpy(), b.cpu().numpy(), outer, C, inner, -8, 7, -128, 127)
    r = O.bwd_pc(gn, xn, s.cpu().numpy(), b.cpu().numpy(), outer, C, inner, -8, 7, -128, 127, True, 1.0, False, numel_for_scaler=n)
    assert torch.equal(y[40:41].cpu(), torch.from_numpy(oy.reshape(xn.shape)).to(torch.bfloat16))
    assert torch.equal(dx[40:41].cpu(), torch.from_numpy(r.dx.reshape(xn.shape)).to(torch.bfloat16))


def test_last_axis_beyond_2_31_elements(dev):
    shape = (266003, 8192)                     # [tokens, features], quantized along the features: 2.18e9 elements, a prime row count
    n = shape[0] * shape[1]
    assert n > (1 << 31)
    x, g = _fill(n, 77, 0.5, 1.0, dev).view(shape), _fill(n, 78, 0.0, 1e-3, dev).view(shape)
    from torchlsq import synth
    s, b = synth.uniform_like(8192, 79, 0.01, 0.05, device=dev), synth.normal_like(8192, 80, 0.0, 0.1, device=dev)
    q = (0, 127, 0, 255, True, 1.0, False, False, False)
    ops = torch.ops.torchlsq_native
    y = ops.lsq_forward_per_channel(x, s, b, 1, *q)
    dx, wide = ops.lsq_backward_per_channel_wide(g, x, s, b, 1, *q, n)
    parts = []
    for lo, hi in ((0, 133000), (133000, 266003)):
        assert torch.equal(ops.lsq_forward_per_channel(x[lo:hi], s, b, 1, *q), y[lo:hi]), "y differs in rows [%d, %d)" % (lo, hi)
        da, wa = ops.lsq_backward_per_channel_wide(g[lo:hi], x[lo:hi], s, b, 1, *q, n)
        assert torch.equal(da, dx[lo:hi]), "dx differs in rows [%d, %d)" % (lo, hi)
        parts.append(wa)
        del da
    np.testing.assert_allclose((parts[0] + parts[1]).cpu().numpy(), wide.cpu().numpy(), rtol=1e-6, atol=1e-12)
    lo = 266003 - 100                          # the last hundred rows against the oracle
    xn, gn = x[lo:].float().cpu().numpy(), g[lo:].float().cpu().numpy()
    oy = O.fwd_pc(xn, s.cpu().numpy(), b.cpu().numpy(), 100, 8192, 1, 0, 127, 0, 255)
    r = O.bwd_pc(gn, xn, s.cpu().numpy(), b.cpu().numpy(), 100, 8192, 1, 0, 127, 0, 255, True, 1.0, False, numel_for_scaler=n)
    assert torch.equal(y[lo:].cpu(), torch.from_numpy(oy.reshape(xn.shape)).to(torch.bfloat16))
    assert torch.equal(dx[lo:].cpu(), torch.from_numpy(r.dx.reshape(xn.shape)).to(torch.bfloat16))
