"""ctypes/numpy front end of the CPU oracle (oracle/lsq_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker / reported baseline.  The product package never does.

All functions take and return numpy arrays in dense memory order.  The per-channel entry points
take the tensor as the 3-D view [outer, C, inner] that reference lsq_cpu.cpp:168-176 builds with
its broadcast scale/shift views.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = [os.path.join(_HERE, "lsq_oracle.c"), os.path.join(_HERE, "lsq_oracle_impl.h")]
_SO = os.path.join(_HERE, "liblsq_oracle.so")


def build(force=False, verbose=False):
    """gcc the plain-C restatement into oracle/liblsq_oracle.so (seconds)."""
    if (not force and os.path.isfile(_SO)
            and all(os.path.getmtime(s) <= os.path.getmtime(_SO) for s in _SRC)):
        return _SO
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp", _SRC[0], "-lm", "-o", _SO]
    if verbose:
        print("[oracle]", " ".join(cmd))
    subprocess.check_call(cmd)
    return _SO


_lib = None
_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_dbl = ctypes.c_double
_vp = ctypes.c_void_p


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        for suf, cT in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
            f = getattr(_lib, "lsq_oracle_grad_scaler_pt_" + suf)
            f.restype = cT
            f.argtypes = [_c_i64, _c_int, _c_int, _c_dbl]
            f = getattr(_lib, "lsq_oracle_grad_scaler_pc_" + suf)
            f.restype = cT
            f.argtypes = [_c_i64, _c_int, _c_i64, _c_int, _c_dbl]
            f = getattr(_lib, "lsq_oracle_fwd_pt_" + suf)
            f.restype = None
            f.argtypes = [_vp, _vp, _c_i64, cT, cT, _c_int, _c_int, _c_int, _c_int, _c_int]
            f = getattr(_lib, "lsq_oracle_levels_pt_" + suf)
            f.restype = None
            f.argtypes = [_vp, _vp, _c_i64, cT, cT, _c_int, _c_int, _c_int, _c_int]
            f = getattr(_lib, "lsq_oracle_bwd_pt_" + suf)
            f.restype = None
            f.argtypes = [_vp, _vp, _vp, _vp, _vp, _c_i64, cT, cT, _c_int, _c_int, _c_int, _c_int,
                          _c_int, _c_dbl, _c_i64, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp]
            f = getattr(_lib, "lsq_oracle_fwd_pc_" + suf)
            f.restype = None
            f.argtypes = [_vp, _vp, _c_i64, _c_i64, _c_i64, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int]
            f = getattr(_lib, "lsq_oracle_levels_pc_" + suf)
            f.restype = None
            f.argtypes = [_vp, _vp, _c_i64, _c_i64, _c_i64, _vp, _vp, _c_int, _c_int, _c_int, _c_int]
            f = getattr(_lib, "lsq_oracle_bwd_pc_" + suf)
            f.restype = None
            f.argtypes = [_vp, _vp, _vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _vp, _vp, _c_int, _c_int,
                          _c_int, _c_int, _c_int, _c_dbl, _c_i64, _c_int, _c_int, _c_int,
                          _vp, _vp, _vp, _vp, _vp, _vp]
        _lib.lsq_oracle_max_threads.restype = _c_int
        _lib.lsq_oracle_set_threads.argtypes = [_c_int]
    return _lib


def max_threads():
    return int(lib().lsq_oracle_max_threads())


def set_threads(n):
    lib().lsq_oracle_set_threads(int(n))


def _suf(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError("oracle supports float32/float64 only (reference lsq_cpu.cpp:37 AT_DISPATCH_FLOATING_TYPES)")


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _dense(a, dtype=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def grad_scaler_pt(numel, quant_max, use_grad_scaling=True, grad_scaler=1.0, dtype=np.float32):
    return float(getattr(lib(), "lsq_oracle_grad_scaler_pt_" + _suf(dtype))(
        int(numel), int(quant_max), int(bool(use_grad_scaling)), float(grad_scaler)))


def grad_scaler_pc(numel, quant_max, channels, use_grad_scaling=True, grad_scaler=1.0, dtype=np.float32):
    return float(getattr(lib(), "lsq_oracle_grad_scaler_pc_" + _suf(dtype))(
        int(numel), int(quant_max), int(channels), int(bool(use_grad_scaling)), float(grad_scaler)))


def fwd_pt(x, scale0, shift0, quant_min, quant_max, type_min, type_max, init_mode=False):
    x = _dense(x)
    y = np.empty_like(x)
    getattr(lib(), "lsq_oracle_fwd_pt_" + _suf(x.dtype))(
        _p(x), _p(y), x.size, float(scale0), float(shift0), quant_min, quant_max, type_min, type_max,
        int(bool(init_mode)))
    return y


def levels_pt(x, scale0, shift0, quant_min, quant_max, type_min, type_max):
    x = _dense(x)
    q = np.empty(x.shape, dtype=np.int32)
    getattr(lib(), "lsq_oracle_levels_pt_" + _suf(x.dtype))(
        _p(x), _p(q), x.size, float(scale0), float(shift0), quant_min, quant_max, type_min, type_max)
    return q


class BwdResult(object):
    """dx plus the reductions; *_wide are fp64 sums of the T-typed terms, abs_* are sum|term|."""
    __slots__ = ("dx", "ds", "db", "ds_wide", "db_wide", "abs_ds", "abs_db", "ds_buf", "db_buf")


def bwd_pt(g, x, scale0, shift0, quant_min, quant_max, type_min, type_max, use_grad_scaling=True,
           grad_scaler=1.0, sym=False, eval_mode=False, init_mode=False, numel_for_scaler=None,
           want_buffers=False):
    x = _dense(x)
    g = _dense(g, x.dtype)
    r = BwdResult()
    r.dx = np.empty_like(x)
    r.ds = np.empty(1, x.dtype)
    r.db = np.empty(1, x.dtype)
    r.ds_wide = np.empty(1, np.float64)
    r.db_wide = np.empty(1, np.float64)
    r.abs_ds = np.empty(1, np.float64)
    r.abs_db = np.empty(1, np.float64)
    r.ds_buf = np.empty_like(x) if want_buffers else None
    r.db_buf = np.empty_like(x) if want_buffers else None
    n4s = x.size if numel_for_scaler is None else int(numel_for_scaler)
    getattr(lib(), "lsq_oracle_bwd_pt_" + _suf(x.dtype))(
        _p(g), _p(x), _p(r.dx), _p(r.ds_buf), _p(r.db_buf), x.size, float(scale0), float(shift0),
        quant_min, quant_max, type_min, type_max, int(bool(use_grad_scaling)), float(grad_scaler), n4s,
        int(bool(sym)), int(bool(eval_mode)), int(bool(init_mode)),
        _p(r.ds), _p(r.db), _p(r.ds_wide), _p(r.db_wide), _p(r.abs_ds), _p(r.abs_db))
    return r


def _ocl(x, outer, C, inner):
    assert x.size == outer * C * inner, "x is not [outer, C, inner]"


def fwd_pc(x, scale, shift, outer, C, inner, quant_min, quant_max, type_min, type_max, init_mode=False):
    x = _dense(x)
    _ocl(x, outer, C, inner)
    scale = _dense(scale, x.dtype)
    shift = _dense(shift, x.dtype)
    assert scale.size == C and shift.size == C
    y = np.empty_like(x)
    getattr(lib(), "lsq_oracle_fwd_pc_" + _suf(x.dtype))(
        _p(x), _p(y), outer, C, inner, _p(scale), _p(shift), quant_min, quant_max, type_min, type_max,
        int(bool(init_mode)))
    return y


def levels_pc(x, scale, shift, outer, C, inner, quant_min, quant_max, type_min, type_max):
    x = _dense(x)
    _ocl(x, outer, C, inner)
    scale = _dense(scale, x.dtype)
    shift = _dense(shift, x.dtype)
    q = np.empty(x.shape, dtype=np.int32)
    getattr(lib(), "lsq_oracle_levels_pc_" + _suf(x.dtype))(
        _p(x), _p(q), outer, C, inner, _p(scale), _p(shift), quant_min, quant_max, type_min, type_max)
    return q


def bwd_pc(g, x, scale, shift, outer, C, inner, quant_min, quant_max, type_min, type_max,
           use_grad_scaling=True, grad_scaler=1.0, sym=False, eval_mode=False, init_mode=False,
           numel_for_scaler=None, want_buffers=False):
    x = _dense(x)
    _ocl(x, outer, C, inner)
    g = _dense(g, x.dtype)
    scale = _dense(scale, x.dtype)
    shift = _dense(shift, x.dtype)
    assert scale.size == C and shift.size == C
    r = BwdResult()
    r.dx = np.empty_like(x)
    r.ds = np.empty(C, x.dtype)
    r.db = np.empty(C, x.dtype)
    r.ds_wide = np.empty(C, np.float64)
    r.db_wide = np.empty(C, np.float64)
    r.abs_ds = np.empty(C, np.float64)
    r.abs_db = np.empty(C, np.float64)
    r.ds_buf = np.empty_like(x) if want_buffers else None
    r.db_buf = np.empty_like(x) if want_buffers else None
    n4s = x.size if numel_for_scaler is None else int(numel_for_scaler)
    getattr(lib(), "lsq_oracle_bwd_pc_" + _suf(x.dtype))(
        _p(g), _p(x), _p(r.dx), _p(r.ds_buf), _p(r.db_buf), outer, C, inner, _p(scale), _p(shift),
        quant_min, quant_max, type_min, type_max, int(bool(use_grad_scaling)), float(grad_scaler), n4s,
        int(bool(sym)), int(bool(eval_mode)), int(bool(init_mode)),
        _p(r.ds), _p(r.db), _p(r.ds_wide), _p(r.db_wide), _p(r.abs_ds), _p(r.abs_db))
    return r


def axis_to_ocl(shape, axis):
    """[outer, C, inner] of a dense row-major tensor quantised along `axis`."""
    outer = int(np.prod(shape[:axis], dtype=np.int64)) if axis > 0 else 1
    inner = int(np.prod(shape[axis + 1:], dtype=np.int64)) if axis + 1 < len(shape) else 1
    return outer, int(shape[axis]), inner


def meanstd(x, outer, C, inner):
    """Per-channel mean and unbiased standard deviation over the [outer, C, inner] view: the statistics of the
    reference module's 3-sigma weight initialisation (quantized/modules/observers.py:329-337: torch.mean / torch.std
    over the non-channel axes, or over everything for C == 1).  The algorithm itself lives in ATen (torch.std =
    sqrt(sum((x - mean)^2) / (n - 1))); restated here as the textbook two-pass form in fp64 -- pinned against
    torch.mean / torch.std and against the scales the reference module produced (tests/golden/module_traces.json)
    in tests/test_oracle_pinned.py.  n == 1 gives std = NaN, like torch."""
    v = np.asarray(x, dtype=np.float64).reshape(outer, C, inner)
    n = outer * inner
    with np.errstate(invalid="ignore", divide="ignore"):
        mu = v.sum(axis=(0, 2)) / n
        dev = v - mu.reshape(1, C, 1)
        var = (dev * dev).sum(axis=(0, 2)) / (n - 1) if n > 1 else np.full(C, np.nan)
        return mu, np.sqrt(var)


def sigma_init_scale(mu, sigma, quant_min, quant_max):
    """scale = max(|mu - 3 sigma|, |mu + 3 sigma|) / 2^bits, bits = ceil(log2(quant_max - quant_min)) - 1
    (reference observers.py:329-337)."""
    import math
    bits = math.ceil(math.log(quant_max - quant_min) / math.log(2)) - 1
    return np.maximum(np.abs(mu - 3 * sigma), np.abs(mu + 3 * sigma)) / 2 ** bits
