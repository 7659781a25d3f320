"""Seeded random sweep of the op surface on the GPU against the CPU oracle (run with -m gpu).

Every case draws a shape (1-5 dims, sizes that cross the packet / wave / window boundaries), a channel axis, a
quantization range, parameter values (negative, zero and tiny scales included), modes, a memory layout and a host
layer, and checks the bars of the parity tests: y, dx bit-exact; d_scale / d_shift within 1e-6 of sum|terms|.
The draws are deterministic (numpy Generator with fixed seeds), so a failure names its case.
"""
import numpy as np
import pytest
import torch

from helpers import assert_bits_equal, assert_reduction_close
from oracle import lsq_oracle as O

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 49, 63, 64, 65, 100, 127, 129, 255, 257, 1000, 1025]
RANGES = [(0, 127, 0, 255), (-64, 63, -128, 127), (-128, 127, -128, 127), (0, 255, 0, 255), (-8, 7, -128, 127), (0, 15, 0, 255),
          (0, 1, 0, 255), (-1, 1, -128, 127), (0, 3, 0, 255)]


def _draw_shape(rng):
    nd = int(rng.integers(1, 6))
    while True:
        shape = tuple(int(rng.choice(SIZES)) for _ in range(nd))
        if int(np.prod(shape)) <= 600_000:
            return shape


def _layout(rng, t, kind):
    """The same values in another memory layout."""
    if kind == "permuted" and t.dim() >= 2:
        perm = list(rng.permutation(t.dim()))
        inv = [perm.index(i) for i in range(t.dim())]
        return t.permute(perm).contiguous().permute(inv)
    if kind == "channels_last" and t.dim() == 4:
        return t.contiguous(memory_format=torch.channels_last)
    if kind == "non_dense" and t.shape[-1] > 0:
        wide = torch.empty(t.shape[:-1] + (2 * t.shape[-1],), dtype=t.dtype, device=t.device)[..., ::2]
        wide.copy_(t)
        return wide
    if kind == "offset":                       # a view starting one element into its storage: 4/8-byte aligned only
        flat = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)[1:]
        flat.copy_(t.reshape(-1))
        return flat.view(t.shape)
    return t


@pytest.mark.parametrize("seed", range(12))
def test_random_cases_against_the_oracle(seed):
    assert torch.cuda.is_available()
    import torchlsq  # noqa: F401
    from torchlsq import extension
    from torchlsq.functional import lsq
    extension._assert_has_ops()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1000 + seed)
    ran = 0
    for case in range(30):
        shape = _draw_shape(rng)
        n = int(np.prod(shape))
        dtype = [torch.float32, torch.float32, torch.float64, torch.bfloat16, torch.float16][int(rng.integers(0, 5))]
        npdt = np.float64 if dtype == torch.float64 else np.float32     # arithmetic / parameter type
        narrow = dtype in (torch.bfloat16, torch.float16)               # 16-bit storage, fp32 math, RNE on store
        per_channel = rng.random() < 0.6
        axis = int(rng.integers(0, len(shape)))
        C = shape[axis] if per_channel else 1
        qmin, qmax, tmin, tmax = RANGES[int(rng.integers(0, len(RANGES)))]
        affine = bool(rng.random() < 0.6) or not (qmin <= 0 <= qmax)
        eval_mode = bool(rng.random() < 0.2)
        init_mode = bool(rng.random() < 0.15)
        use_gs = bool(rng.random() < 0.8)
        gs = float(rng.choice([1.0, 0.5, 3.0]))
        step = float(rng.choice([0.003, 0.05, 0.4, 2.0]))
        x = (rng.standard_normal(n) * step * (qmax - qmin) * 0.4 + step * (qmax + qmin) * 0.5).astype(npdt)
        x[rng.integers(0, n, size=min(n, 8))] = rng.choice(np.array([0.0, step * qmin, step * qmax, step * (qmin - 0.5),
                                                                     step * (qmax + 0.5), step * 0.5, -step * 0.5], dtype=npdt), size=min(n, 8))
        g = (rng.standard_normal(n) * 1e-2).astype(npdt)
        if narrow:      # the stored values ARE the inputs: round them to the storage type first
            x = torch.from_numpy(x).to(dtype).to(torch.float32).numpy()
            g = torch.from_numpy(g).to(dtype).to(torch.float32).numpy()
        scale = (rng.uniform(0.5, 1.5, size=C) * step).astype(npdt)
        if rng.random() < 0.3:
            scale[int(rng.integers(0, C))] *= -1.0            # |scale| is used (lsq_cpu.cpp:45)
        if rng.random() < 0.1:
            scale[int(rng.integers(0, C))] = 0.0              # clamped to eps
        shift = (rng.standard_normal(C) * step * (2.0 if affine else 0.0)).astype(npdt)
        kind = str(rng.choice(["contiguous", "contiguous", "permuted", "channels_last", "non_dense", "offset"]))
        binding = "native" if rng.random() < 0.5 else "ctypes"
        route = str(rng.choice(["functional", "functional", "dispatcher"]))
        tag = "seed %d case %d: %s %s pc=%s axis=%d q=(%d,%d,%d,%d) affine=%s eval=%s init=%s gs=(%s,%s) %s %s/%s" % (
            seed, case, shape, dtype, per_channel, axis, qmin, qmax, tmin, tmax, affine, eval_mode, init_mode, use_gs, gs, kind,
            binding, route)

        xs = x.reshape(shape)
        if per_channel:
            outer, C_, inner = O.axis_to_ocl(shape, axis)
            oy = O.fwd_pc(xs, scale, shift, outer, C_, inner, qmin, qmax, tmin, tmax, init_mode)
            r = O.bwd_pc(g.reshape(shape), xs, scale, shift, outer, C_, inner, qmin, qmax, tmin, tmax, use_gs, gs, not affine,
                         eval_mode, init_mode)
        else:
            oy = O.fwd_pt(xs, scale[0], shift[0], qmin, qmax, tmin, tmax, init_mode)
            r = O.bwd_pt(g.reshape(shape), xs, scale[0], shift[0], qmin, qmax, tmin, tmax, use_gs, gs, not affine, eval_mode,
                         init_mode)

        extension.set_host_binding(binding)
        try:
            xt = _layout(rng, torch.from_numpy(xs).to(dev).to(dtype), kind).requires_grad_(True)
            gt = torch.from_numpy(g.reshape(shape)).to(dev).to(dtype)
            st = torch.from_numpy(scale).to(dev).requires_grad_(True)
            bt = torch.from_numpy(shift).to(dev).requires_grad_(True)
            args = (qmin, qmax, tmin, tmax, axis, use_gs, gs, affine, per_channel, eval_mode, init_mode)
            y = lsq(xt, st, bt, *args) if route == "functional" else torch.ops.torchlsq.lsq(xt, st, bt, *args)
            y.backward(gt)
            torch.cuda.synchronize()
        finally:
            extension.set_host_binding("native")
        if narrow:      # parity for 16-bit storage is defined by the build: the fp32 result rounded to the storage type
            want_y = torch.from_numpy(np.ascontiguousarray(oy)).to(dtype)
            want_dx = torch.from_numpy(np.ascontiguousarray(r.dx)).to(dtype)
            assert torch.equal(y.detach().cpu().view(torch.int16), want_y.view(torch.int16)), tag + " y"
            assert torch.equal(xt.grad.cpu().view(torch.int16), want_dx.view(torch.int16)), tag + " dx"
        else:
            assert_bits_equal(y.detach().cpu().numpy(), oy, tag + " y")
            assert_bits_equal(xt.grad.cpu().numpy(), r.dx, tag + " dx")
        ds = st.grad.cpu().numpy() if st.grad is not None else np.zeros(C, npdt)
        db = bt.grad.cpu().numpy() if bt.grad is not None else np.zeros(C, npdt)
        assert_reduction_close(ds, r.ds_wide, r.abs_ds, tag + " ds")
        assert_reduction_close(db, r.db_wide, r.abs_db, tag + " db")
        ran += 1
    assert ran == 30
