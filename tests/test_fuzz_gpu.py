"""Seeded random sweep of the op surface on the GPU against the CPU oracle (run with -m gpu).

Every case draws a shape (1-5 dims, sizes that cross the packet / wave / window boundaries), a channel axis, a
quantization range, parameter values (negative, zero and tiny scales included), modes, a memory layout and a host
layer, and checks the bars of the parity tests: y, dx bit-exact; d_scale / d_shift within 1e-6 of sum|terms|.
The draws are deterministic (numpy Generator with fixed seeds), so a failure names its case.
"""
import os

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


N_SEEDS = int(os.environ.get("LSQ_FUZZ_SEEDS", "12"))      # a longer soak: LSQ_FUZZ_SEEDS=200 python -m pytest tests/test_fuzz_gpu.py -m gpu


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_cases_against_the_oracle(seed):
    assert torch.cuda.is_available()
    import torchlsq  # noqa: F401
    import lsq_tools
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

        # the window-mode per-channel kernels have two loop forms (register loops, LDS-DMA ring) and the launch policy
        # picks by shape: force either one on a share of the cases so that both meet every kind of input
        loop_form = int(rng.choice([0, 0, 1, 2, 2]))
        # ... and the row-group windows (quantized axis last) two workgroup sizes: 768/1024 lanes on a share of the cases
        ww_big = int(rng.choice([0, 0, 1]))
        # ... and the ring's copies with or without the streaming hint (the policy uses it above 32 MB only)
        ring_nt = int(rng.choice([0, 1, 2]))
        # The knobs exist only in the TOOLS build of the library (tools/lsq_tools.py), which the ctypes host layer can be
        # pointed at; the C++ binding is linked against the production library, so its cases run the built-in policy.
        if binding == "native":
            loop_form = ww_big = ring_nt = 0
        forced = bool(loop_form or ww_big or ring_nt)
        tag += " loop=%d big=%d nt=%d" % (loop_form, ww_big, ring_nt)
        try:
            if forced:
                tl = lsq_tools.activate()
                tl.lsq_hip_debug_force_ring(loop_form)
                tl.lsq_hip_debug_set_ww_big(ww_big)
                tl.lsq_hip_debug_set_ring_nt(ring_nt)
            else:
                extension.set_host_binding(binding)
            xt = _layout(rng, torch.from_numpy(xs).to(dev).to(dtype), kind).requires_grad_(True)
            gt = torch.from_numpy(g.reshape(shape)).to(dev).to(dtype)
            st = torch.from_numpy(scale).to(dev).requires_grad_(True)
            bt = torch.from_numpy(shift).to(dev).requires_grad_(True)
            args = (qmin, qmax, tmin, tmax, axis, use_gs, gs, affine, per_channel, eval_mode, init_mode)
            y = lsq(xt, st, bt, *args) if route == "functional" else torch.ops.torchlsq.lsq(xt, st, bt, *args)
            y.backward(gt)
            torch.cuda.synchronize()
        finally:
            if forced:
                lsq_tools.deactivate()
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


@pytest.mark.parametrize("seed", range(max(6, N_SEEDS // 2)))
def test_random_cases_of_the_side_outputs(seed):
    """Same draws for the ops beside the reference's four: integer levels (int8, bit-exact against the oracle's
    rne(clamp(x/s + zp))), the eval-mode mask + backward_from_mask, one-pass min/max (exact) and mean/std (1e-6)."""
    assert torch.cuda.is_available()
    import torchlsq  # noqa: F401
    from torchlsq import extension
    extension._assert_has_ops()
    ops = torch.ops.torchlsq
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5000 + seed)
    for case in range(25):
        shape = _draw_shape(rng)
        n = int(np.prod(shape))
        dtype = [torch.float32, torch.float64][int(rng.integers(0, 2))]
        npdt = np.float64 if dtype == torch.float64 else np.float32
        per_channel = rng.random() < 0.6
        axis = int(rng.integers(0, len(shape)))
        C = shape[axis] if per_channel else 1
        qmin, qmax, tmin, tmax = RANGES[int(rng.integers(0, len(RANGES)))]
        step = float(rng.choice([0.003, 0.05, 0.4, 2.0]))
        x = (rng.standard_normal(n) * step * (qmax - qmin) * 0.4 + step * (qmax + qmin) * 0.5).astype(npdt).reshape(shape)
        g = (rng.standard_normal(n) * 1e-2).astype(npdt).reshape(shape)
        scale = (rng.uniform(0.5, 1.5, size=C) * step).astype(npdt)
        shift = (rng.standard_normal(C) * step * 2.0).astype(npdt)
        kind = str(rng.choice(["contiguous", "permuted", "channels_last", "non_dense", "offset"]))
        tag = "seed %d case %d: %s %s pc=%s axis=%d q=(%d,%d,%d,%d) %s" % (seed, case, shape, dtype, per_channel, axis, qmin, qmax,
                                                                         tmin, tmax, kind)
        xt = _layout(rng, torch.from_numpy(x).to(dev), kind)
        gt = torch.from_numpy(g).to(dev)
        st, bt = torch.from_numpy(scale).to(dev), torch.from_numpy(shift).to(dev)
        bias = 128 if qmax > 127 else 0
        if per_channel:
            outer, C_, inner = O.axis_to_ocl(shape, axis)
            want_q = O.levels_pc(x, scale, shift, outer, C_, inner, qmin, qmax, tmin, tmax)
            want_y = O.fwd_pc(x, scale, shift, outer, C_, inner, qmin, qmax, tmin, tmax)
            r = O.bwd_pc(g, x, scale, shift, outer, C_, inner, qmin, qmax, tmin, tmax, True, 1.0, False, True, False)
            y, q = ops.lsq_quantize_per_channel(xt, st, bt, axis, qmin, qmax, tmin, tmax, bias)
            y2, mask = extension.hip_forward_per_channel(xt, st, bt, axis, qmin, qmax, tmin, tmax, True, 1.0, False, True, False,
                                                         want_mask=True)
            mn, mx = ops.lsq_minmax_per_channel(xt, axis)
            mu, sd = ops.lsq_meanstd_per_channel(xt, axis)
            moved = np.moveaxis(x, axis, 0).reshape(C, -1).astype(np.float64)
            want_mu, want_sd = O.meanstd(x, outer, C_, inner)
        else:
            want_q = O.levels_pt(x, scale[0], shift[0], qmin, qmax, tmin, tmax)
            want_y = O.fwd_pt(x, scale[0], shift[0], qmin, qmax, tmin, tmax)
            r = O.bwd_pt(g, x, scale[0], shift[0], qmin, qmax, tmin, tmax, True, 1.0, False, True, False)
            y, q = ops.lsq_quantize_per_tensor(xt, st, bt, qmin, qmax, tmin, tmax, bias)
            y2, mask = extension.hip_forward_per_tensor(xt, st, bt, qmin, qmax, tmin, tmax, True, 1.0, False, True, False,
                                                        want_mask=True)
            mn, mx = ops.lsq_minmax_per_tensor(xt)
            mu, sd = ops.lsq_meanstd_per_tensor(xt)
            moved = x.reshape(1, -1).astype(np.float64)
            want_mu, want_sd = O.meanstd(x, 1, 1, n)
        assert q.dtype == torch.int8 and q.shape == xt.shape
        assert np.array_equal(q.cpu().numpy().astype(np.int32) + bias, want_q.reshape(shape)), tag + " levels"
        assert_bits_equal(y.cpu().numpy(), want_y, tag + " y (quantize op)")
        assert_bits_equal(y2.cpu().numpy(), want_y, tag + " y (masked forward)")
        dx = ops.lsq_backward_from_mask(gt, mask)
        assert_bits_equal(dx.cpu().numpy(), r.dx, tag + " dx from mask")
        assert np.array_equal(mn.cpu().numpy().reshape(-1).astype(np.float64), moved.min(axis=1)), tag + " min"
        assert np.array_equal(mx.cpu().numpy().reshape(-1).astype(np.float64), moved.max(axis=1)), tag + " max"
        rtol = 1e-12 if dtype == torch.float64 else 1e-6
        spread = float(np.abs(moved).max()) + 1e-30
        np.testing.assert_allclose(mu.cpu().numpy().reshape(-1), want_mu, rtol=rtol, atol=rtol * spread, err_msg=tag + " mean")
        np.testing.assert_allclose(sd.cpu().numpy().reshape(-1), want_sd, rtol=rtol, atol=0, equal_nan=True, err_msg=tag + " std")
