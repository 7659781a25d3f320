"""Python host layer over liblsq_cpu.so (include/lsq_cpu.h): the kernels for tensors in host memory, registered by
`extension.py` under the "CPU" dispatch key -- the counterpart of the reference's TORCH_LIBRARY_IMPL(torchlsq, CPU)
(lsq_cpu.cpp:298-311).  Same checks, layout handling and outputs as _hip_host.py; never reached by a GPU tensor.
"""
import ctypes  # noqa: F401

import torch

from . import _abi
from ._abi import _DTYPE_CODE, _assert_has_ops
from ._hip_host import (_check, _dense, _like_layout, _ocl, _param_dtype, _params, _require_param,  # noqa: F401
                        check_backward_dtypes, check_channel_args, check_forward_dtypes)

# -------------------------------------------------------------------------------------------------
# the CPU backend ("CPU" dispatch key): host-memory tensors -> liblsq_cpu.so, the counterpart of the reference's
# TORCH_LIBRARY_IMPL(torchlsq, CPU) (lsq_cpu.cpp:298-311).  Same checks, same layout handling, same outputs as the HIP
# backend above; never reached by a GPU tensor.
# -------------------------------------------------------------------------------------------------
def _cpu_lib(what):
    _assert_has_ops()      # the package as a whole needs its HIP library: CPU tensors do not make it usable on their own
    if _abi._CPU_LIB is None:
        raise NotImplementedError("%s: the CPU kernels (liblsq_cpu.so) are not available: %s" % (what, _abi.cpu_error_str))
    _abi._CPU_LIB.lsq_cpu_set_num_threads(torch.get_num_threads())     # the loops follow torch's intra-op thread setting
    return _abi._CPU_LIB


def _require_cpu(what, *tensors):
    for t in tensors:
        if t.device.type != "cpu":
            raise RuntimeError("%s: expected all tensors on the CPU but got one on %s" % (what, t.device))


def _cpu_status(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, _abi._CPU_LIB.lsq_cpu_last_error().decode("utf-8", "replace")))


def _cpu_dtype(x, what):
    _check(x.dtype in (torch.float32, torch.float64, torch.bfloat16),
           '"%s" not implemented for \'%s\'' % (what, str(x.dtype).replace("torch.", "")))
    return _DTYPE_CODE[x.dtype]


def cpu_forward(x, scale, shift, axis, per_channel, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode):
    what = "lsq_forward_per_channel" if per_channel else "lsq_forward_per_tensor"
    lib = _cpu_lib(what)
    code = _cpu_dtype(x, "lsq_forward")
    check_forward_dtypes(x, scale, shift)
    if per_channel:
        check_channel_args(x, scale, shift, axis, backward=False)
    _require_cpu(what, x, scale, shift)
    xd, order = _dense(x)
    y = torch.empty_like(xd)
    if xd.numel() == 0:
        return y
    _require_param(what, scale, shift)
    _, pref = _params(qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode)
    sc, sh = scale.contiguous(), shift.contiguous()
    if per_channel:
        outer, C, inner = _ocl(xd, order, axis)
        rc = lib.lsq_cpu_forward_per_channel(code, xd.data_ptr(), y.data_ptr(), outer, C, inner, sc.data_ptr(), sh.data_ptr(), pref)
    else:
        rc = lib.lsq_cpu_forward_per_tensor(code, xd.data_ptr(), y.data_ptr(), xd.numel(), sc.data_ptr(), sh.data_ptr(), pref)
    _cpu_status(rc, what)
    return y


def cpu_backward(grad, x, scale, shift, axis, per_channel, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode,
                 numel_for_scaler=0, want_wide=False, wide_out=None):
    what = "lsq_backward_per_channel" if per_channel else "lsq_backward_per_tensor"
    lib = _cpu_lib(what)
    code = _cpu_dtype(x, "lsq_backward")
    check_backward_dtypes(grad, x, scale, shift)
    if per_channel:
        check_channel_args(x, scale, shift, axis, backward=True)
    C = scale.numel() if per_channel else 1
    if x.numel() <= 0:  # lsq_cpu.cpp:76-78, :221-223 return (x, scale, shift) themselves
        if want_wide:
            if wide_out is not None:
                wide_out[:2 * C].zero_()
                return x.clone(), wide_out
            return x.clone(), torch.zeros((2, C) if per_channel else (2,), dtype=torch.float64)
        return x.clone(), scale.clone(), shift.clone()
    _require_cpu(what, x, grad, scale, shift)
    _require_param(what, scale, shift)
    xd, order = _dense(x)
    gd = _like_layout(grad, xd)
    dx = torch.empty_like(xd)
    pd = _param_dtype(x)
    ds, db = torch.empty(C, dtype=pd), torch.empty(C, dtype=pd)
    wide = None
    if want_wide and wide_out is not None:      # the caller's buffer (the sharded path packs the element count behind the sums)
        _check(wide_out.dtype == torch.float64 and wide_out.is_contiguous() and wide_out.numel() >= 2 * C and
               wide_out.device.type == "cpu", "wide_out must be a contiguous float64 CPU tensor of at least 2 * channels elements")
        wide = wide_out
    elif want_wide:
        wide = torch.empty((2, C) if per_channel else (2,), dtype=torch.float64)
    _, pref = _params(qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode, numel_for_scaler)
    sc, sh = scale.contiguous(), shift.contiguous()
    wptr = wide.data_ptr() if want_wide else None
    if per_channel:
        outer, C_, inner = _ocl(xd, order, axis)
        rc = lib.lsq_cpu_backward_per_channel(code, gd.data_ptr(), xd.data_ptr(), dx.data_ptr(), ds.data_ptr(), db.data_ptr(),
                                              wptr, outer, C_, inner, sc.data_ptr(), sh.data_ptr(), pref)
    else:
        rc = lib.lsq_cpu_backward_per_tensor(code, gd.data_ptr(), xd.data_ptr(), dx.data_ptr(), ds.data_ptr(), db.data_ptr(),
                                             wptr, xd.numel(), sc.data_ptr(), sh.data_ptr(), pref)
    _cpu_status(rc, what)
    if want_wide:
        return dx, wide
    return dx, ds, db


def cpu_levels(x, scale, shift, axis, per_channel, qmin, qmax, tmin, tmax, level_bias):
    """The integer levels of lsq_kernel.h:13 for a tensor in host memory, as the byte (q - level_bias) mod 256: the formula of
    the forward kernels operation by operation in torch's own (individually rounded) tensor arithmetic -- a conversion-time
    helper on the CPU, not a hot path."""
    _cpu_lib("lsq_levels")
    check_forward_dtypes(x, scale, shift)
    _require_cpu("lsq_levels", x, scale, shift)
    if per_channel:
        check_channel_args(x, scale, shift, axis, backward=False)
    lo, hi = qmin - level_bias, qmax - level_bias
    _check((lo >= -128 and hi <= 127) or (lo >= 0 and hi <= 255),
           "levels: [quant_min, quant_max] - level_bias = [%d, %d] fits neither int8 nor uint8" % (lo, hi))
    t = _param_dtype(x)
    xf = x.detach().to(t)
    eps = torch.finfo(t).eps
    s = scale.detach().abs().clamp_min(eps)
    inv_s = 1.0 / s
    zp = torch.fmin(torch.full_like(s, tmax), torch.fmax(torch.full_like(s, tmin), -shift.detach() * inv_s)).round()
    if per_channel:
        view = [1] * x.dim()
        view[axis] = -1
        inv_s, zp = inv_s.view(view), zp.view(view)
    q = xf * inv_s
    q = q + zp
    q = torch.fmin(torch.full_like(q, qmax), torch.fmax(torch.full_like(q, qmin), q)).round()      # NaN -> quant_min, like the kernels
    return (q - level_bias).to(torch.int32).to(torch.int8)


def cpu_sharded_finish(packed, channels, per_channel, x_dtype, qmax, use_gs, gs):
    """lsq_cpu_sharded_finish: (d_scale, d_shift) from the all-reduced [sum ds (C), sum db (C), element count] (see
    _hip_host.hip_sharded_finish)."""
    lib = _cpu_lib("lsq_sharded_finish")
    _require_cpu("lsq_sharded_finish", packed)
    _check(packed.dtype == torch.float64 and packed.is_contiguous() and packed.numel() == 2 * channels + 1,
           "lsq_sharded_finish: packed must be a contiguous float64 tensor of 2 * channels + 1 elements")
    pd = torch.float64 if x_dtype == torch.float64 else torch.float32
    ds, db = torch.empty(channels, dtype=pd), torch.empty(channels, dtype=pd)
    _, pref = _params(0, qmax, 0, qmax, use_gs, gs, False, False, False)
    rc = lib.lsq_cpu_sharded_finish(_DTYPE_CODE[x_dtype], packed.data_ptr(), channels, 1 if per_channel else 0, pref,
                                    ds.data_ptr(), db.data_ptr())
    _cpu_status(rc, "lsq_sharded_finish")
    return ds, db


def _cpu_minmax(x, axis=None):
    """torch's own reductions: the stock observers' arithmetic (reference observers.py:446-449 calls them on CPU tensors)"""
    y = x.detach().to(_param_dtype(x))
    if axis is None:
        return torch.aminmax(y)
    dims = [d for d in range(x.dim()) if d != axis]
    return torch.amin(y, dims), torch.amax(y, dims)


def _cpu_meanstd(x, axis=None):
    y = x.detach().to(_param_dtype(x))
    if axis is None:
        return y.mean(), y.std()
    dims = [d for d in range(x.dim()) if d != axis]
    return torch.mean(y, dims), torch.std(y, dims)

