"""Deterministic synthetic inputs for the LSQ hot path (bench, parity tests, golden fixtures).

A counter-based generator: element i of stream `seed` depends only on (seed, i), is computed
with 64-bit INTEGER arithmetic (splitmix64 finaliser) and turned into an approximately normal
variate WITHOUT transcendental functions (Irwin-Hall: the sum of the four 16-bit fields of the
hashed word), so the very same bits come out of numpy-free torch code on the CPU here and on the
GPU on the MI355X box.  That is what lets tests/golden/ pin BASELINE-sized tensors (205 M
elements) by digest instead of shipping gigabytes.

This module deliberately imports nothing from the rest of the package (tests/golden/make_golden.py
loads it by path, in a process where the reference's own `torchlsq::*` ops are registered).
"""
import math

import torch

_MASK64_LO = (1 << 63) - 1  # python ints are unbounded; torch int64 wraps like uint64 bit patterns

_GOLDEN = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB


def _s64(v):
    """python int (as uint64 bit pattern) -> the int64 value with the same bits."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(z, k):
    """logical shift right of an int64 tensor holding uint64 bit patterns."""
    return (z >> k) & ((1 << (64 - k)) - 1)


def hash64(index, seed):
    """splitmix64 finaliser of (seed * golden + index); int64 tensor in, int64 bit patterns out."""
    z = index + _s64((int(seed) + 1) * _GOLDEN)
    z = (z ^ _lsr(z, 30)) * _s64(_M1)
    z = (z ^ _lsr(z, 27)) * _s64(_M2)
    return z ^ _lsr(z, 31)


_IH_MEAN = 2 * 65535            # mean of the sum of four uniform u16
_IH_STD = math.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0)


def normal_like(numel, seed, mean=0.0, std=1.0, device="cpu", dtype=torch.float32, chunk=1 << 24,
                absolute=False, first_index=0):
    """1-D tensor of `numel` approximately N(mean, std) values (|.| of the variate if `absolute`): elements
    first_index .. first_index + numel - 1 of stream `seed` (a slice of a longer stream is that stream's bits).

    value_i = fp64(mean) + fp64(std / IH_STD) * (sum of the 4 u16 fields of hash64(i) - IH_MEAN),
    each step a single IEEE operation in fp64, then one rounding to `dtype`.
    """
    out = torch.empty(numel, dtype=dtype, device=device)
    k = float(std) / _IH_STD
    for lo in range(0, numel, chunk):
        hi = min(numel, lo + chunk)
        idx = torch.arange(lo + int(first_index), hi + int(first_index), dtype=torch.int64, device=device)
        h = hash64(idx, seed)
        s = (h & 0xFFFF) + (_lsr(h, 16) & 0xFFFF) + (_lsr(h, 32) & 0xFFFF) + _lsr(h, 48)
        v = (s - _IH_MEAN).to(torch.float64)
        if absolute:
            v = v.abs()
        v = v * k
        v = v + float(mean)
        out[lo:hi] = v.to(dtype)
    return out


def uniform_like(numel, seed, low=0.0, high=1.0, device="cpu", dtype=torch.float32):
    """1-D tensor of `numel` uniform values in [low, high): top 24 bits of the hash / 2^24."""
    idx = torch.arange(numel, dtype=torch.int64, device=device)
    u = _lsr(hash64(idx, seed), 40).to(torch.float64) * (1.0 / (1 << 24))
    u = u * (float(high) - float(low))
    u = u + float(low)
    return u.to(dtype)


# ---------------------------------------------------------------------------------------------
# The BASELINE.json configurations (SURVEY.md section 8(d) "Synthetic inputs").
# ---------------------------------------------------------------------------------------------
CONFIGS = {
    # per-tensor quint8 activations; 7-bit default range of LSQFakeQuantizer (observers.py:233-237)
    "cfg1": dict(shape=(4, 64, 56, 56), per_channel=False, axis=1, qmin=0, qmax=127, tmin=0, tmax=255,
                 affine=True, x_mean=1.5, x_std=1.0, scale=0.03, shift=0.0, dtype="float32"),
    "cfg2": dict(shape=(128, 512, 56, 56), per_channel=False, axis=1, qmin=0, qmax=127, tmin=0, tmax=255,
                 affine=True, x_mean=1.5, x_std=1.0, scale=0.03, shift=0.0, dtype="float32"),
    # per-channel qint8 weights, ch_axis=0, symmetric
    "cfg3": dict(shape=(512, 512, 3, 3), per_channel=True, axis=0, qmin=-128, qmax=127, tmin=-128, tmax=127,
                 affine=False, x_mean=0.0, x_std=0.05, scale=(5e-4, 2.5e-3), shift=0.0, dtype="float32"),
    # cfg4 is cfg2's operator on [1024,1024,14,14], batch-sharded over the ranks
    "cfg4": dict(shape=(1024, 1024, 14, 14), per_channel=False, axis=1, qmin=0, qmax=127, tmin=0, tmax=255,
                 affine=True, x_mean=1.5, x_std=1.0, scale=0.03, shift=0.0, dtype="float32"),
    # 4-bit per-channel, axis 1 (activation style); bf16 I/O on the GPU, fp32 stand-in for the oracle
    "cfg5": dict(shape=(256, 2048, 7, 7), per_channel=True, axis=1, qmin=-8, qmax=7, tmin=-128, tmax=127,
                 affine=True, x_mean=0.0, x_std=1.0, scale=(0.05, 0.35), shift=("normal", 0.0, 0.1),
                 dtype="bfloat16"),
    # NOT in BASELINE.json: the token-layout activation shapes the round-1 review singled out (the quantized axis is the
    # last one: row-group windows), per-channel quint8, as bench.py workloads `tok` / `vit` (+ `_bf16`)
    "tok": dict(shape=(8192, 4096), per_channel=True, axis=1, qmin=0, qmax=127, tmin=0, tmax=255,
                affine=True, x_mean=0.5, x_std=1.0, scale=(0.01, 0.05), shift=("normal", 0.0, 0.1), dtype="float32"),
    "vit": dict(shape=(64, 197, 768), per_channel=True, axis=2, qmin=0, qmax=127, tmin=0, tmax=255,
                affine=True, x_mean=0.5, x_std=1.0, scale=(0.01, 0.05), shift=("normal", 0.0, 0.1), dtype="float32"),
    # NOT in BASELINE.json either: NCHW activations inside the band the launch policy serves with OWNER windows (at most 13 M
    # fp32 / 20 M 16-bit elements, short channel rows; lsq_pc_geom.hpp plan_own) -- digest-pinned against the reference so that
    # the kernel family the SHIPPED library picks for them is held to the reference's own per-channel backward
    # (tests/test_shipped_binary_gpu.py).  own33: 33 rows leave a short last row tile (the loop's ragged form).
    "own33": dict(shape=(33, 2048, 7, 7), per_channel=True, axis=1, qmin=-8, qmax=7, tmin=-128, tmax=127,
                  affine=True, x_mean=0.0, x_std=1.0, scale=(0.05, 0.35), shift=("normal", 0.0, 0.1), dtype="float32"),
    "own64": dict(shape=(64, 2048, 7, 7), per_channel=True, axis=1, qmin=-8, qmax=7, tmin=-128, tmax=127,
                  affine=False, x_mean=0.0, x_std=1.0, scale=(0.05, 0.35), shift=0.0, dtype="float32"),
    "own16": dict(shape=(16, 1024, 14, 14), per_channel=True, axis=1, qmin=0, qmax=127, tmin=0, tmax=255,
                  affine=True, x_mean=0.8, x_std=1.0, scale=(0.01, 0.05), shift=("normal", 0.0, 0.1), dtype="float32"),
    "own32": dict(shape=(32, 512, 28, 28), per_channel=True, axis=1, qmin=0, qmax=255, tmin=0, tmax=255,
                  affine=True, x_mean=0.8, x_std=1.0, scale=(0.005, 0.02), shift=("normal", 0.0, 0.1), dtype="float32"),
    # ... and a last-axis activation big enough for the fp32 row groups' LDS-DMA ring (from 2^24 elements on)
    "rgring": dict(shape=(16400, 1024), per_channel=True, axis=1, qmin=0, qmax=127, tmin=0, tmax=255,
                   affine=True, x_mean=0.5, x_std=1.0, scale=(0.01, 0.05), shift=("normal", 0.0, 0.1), dtype="float32"),
}

SEED_X, SEED_G, SEED_SCALE, SEED_SHIFT = 11, 23, 37, 41
GRAD_STD = 1e-3


def ds_term_sign(x, scale, shift, c):
    """+1 / -1 per element: the sign of the factor the d_scale term of the element multiplies its gradient with
    (reference lsq_kernel.h:115-121: (x_r - x) / s inside the range, quant_min - zp or quant_max - zp on a border).

    Single IEEE fp32 operations only, the per-channel constants (a division) always on the CPU, so the CPU and the GPU
    give the same bits.  Used for the `abs_grad="dspos"` gradients below; finite inputs only."""
    f32 = torch.float32
    eps = torch.finfo(f32).eps
    sc, sh = scale.detach().to("cpu", f32), shift.detach().to("cpu", f32)
    s = torch.clamp(sc.abs(), min=eps)
    inv_s = 1.0 / s
    zp = torch.round(torch.clamp(-sh * inv_s, min=float(c["tmin"]), max=float(c["tmax"])))
    if c["per_channel"]:
        view = [1] * x.dim()
        view[c["axis"]] = x.shape[c["axis"]]
        s, inv_s, zp = (t.view(view).to(x.device) for t in (s, inv_s, zp))
    else:
        s, inv_s, zp = (t.to(x.device) for t in (s, inv_s, zp))
    xf = x.detach().to(f32)
    t = xf * inv_s
    t = t + zp
    xq = torch.clamp(t, min=float(c["qmin"]), max=float(c["qmax"]))
    inside = (xq > float(c["qmin"])) & (xq < float(c["qmax"]))
    d = torch.round(xq) - zp
    err = d * s
    err = err - xf
    coef = torch.where(inside, err, d)
    return torch.where(coef < 0, -torch.ones_like(coef), torch.ones_like(coef))


def make_inputs(cfg, device="cpu", dtype=None, shape=None, abs_grad=False, first_index=0):
    """(x, grad, scale, shift) for a CONFIGS entry (or a dict of the same keys).

    `shape` overrides the configured shape (e.g. one rank's shard); `dtype` overrides the storage
    type of x/grad (scale/shift are always fp32 unless dtype is float64).
    `abs_grad`: False -- grad ~ N(0, GRAD_STD), mixed sign; True -- |grad| (every d_shift term and every border term of one
    side has one sign: no cancellation in a per-tensor quint8 d_scale, still some in a signed range); "dspos" -- |grad| times
    the sign of the element's d_scale factor (ds_term_sign): every d_scale term is >= 0, the no-cancellation case of d_scale
    for any range.
    `first_index`: x / grad are elements first_index .. first_index + numel - 1 of their streams -- with `shape` one rank's
    dim-0 slice of the configured tensor (rank r of N holds rows [r * rows / N, (r + 1) * rows / N): first_index = r * numel of
    a shard), the SAME bits the slice of the whole tensor has.  scale / shift are replicated, never offset.
    """
    c = CONFIGS[cfg] if isinstance(cfg, str) else cfg
    shape = tuple(shape if shape is not None else c["shape"])
    numel = 1
    for d in shape:
        numel *= d
    dt = dtype if dtype is not None else getattr(torch, c["dtype"])
    pdt = torch.float64 if dt == torch.float64 else torch.float32
    x = normal_like(numel, SEED_X, c["x_mean"], c["x_std"], device, dt, first_index=first_index).view(shape)
    g = normal_like(numel, SEED_G, 0.0, GRAD_STD, device, dt, absolute=bool(abs_grad), first_index=first_index).view(shape)
    C = shape[c["axis"]] if c["per_channel"] else 1
    sc = c["scale"]
    if isinstance(sc, tuple):
        scale = uniform_like(C, SEED_SCALE, sc[0], sc[1], device, pdt)
    else:
        scale = torch.full((C,), float(sc), dtype=pdt, device=device)
    sh = c["shift"]
    if isinstance(sh, tuple):
        shift = normal_like(C, SEED_SHIFT, sh[1], sh[2], device, pdt)
    else:
        shift = torch.full((C,), float(sh), dtype=pdt, device=device)
    if abs_grad == "dspos":
        g = g * ds_term_sign(x, scale, shift, c).to(g.dtype)
    return x, g, scale, shift


def op_kwargs(cfg):
    """keyword arguments of torchlsq.functional.lsq for a CONFIGS entry (training mode)."""
    c = CONFIGS[cfg] if isinstance(cfg, str) else cfg
    return dict(quant_min=c["qmin"], quant_max=c["qmax"], type_min=c["tmin"], type_max=c["tmax"],
                axis=c["axis"], use_grad_scaling=True, grad_scaler=1.0, is_affine=c["affine"],
                is_perchannel=c["per_channel"], eval_mode=False, init_mode=False)
