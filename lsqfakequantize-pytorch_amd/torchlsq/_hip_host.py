"""Python host layer over the C ABI of liblsq_hip.so: argument checks (the reference's TORCH_CHECKs), memory-layout
handling, cached parameter structs, workspaces, tickets and one ctypes call per op.  `extension.py` registers these
functions as the "CUDA"-key kernels of the torchlsq ops (HIP tensors carry that key on ROCm); the C++ binding
(csrc/torch_binding/lsq_torch_binding.cpp) is the second host layer over the same entry points.  A tensor that is not on a
GPU never gets here, and nothing here computes on the host.
"""
import ctypes
import os
import threading

import torch

from . import _abi
from ._abi import (LSQ_TICKET_BYTES, LsqBwdExtras, LsqFwdExtras, LsqObserverUpdate, LsqParams, LsqPcItem, _DTYPE_CODE,  # noqa: F401
                   _assert_has_ops)

# -------------------------------------------------------------------------------------------------
# argument checks (same conditions and messages as the reference's TORCH_CHECKs)
# -------------------------------------------------------------------------------------------------
def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _param_dtype(x):
    """dtype scale/shift must have for input x (reference: identical to x, lsq_cpu.cpp:28-29).
    Extension: 16-bit inputs take fp32 parameters."""
    return torch.float32 if x.dtype in (torch.bfloat16, torch.float16) else x.dtype


def check_forward_dtypes(x, scale, shift):
    _check(x.dtype in _DTYPE_CODE, '"lsq_forward" not implemented for \'%s\'' % str(x.dtype).replace("torch.", ""))
    pd = _param_dtype(x)
    _check(scale.dtype == pd, "`input` and `scale` must have the same floating-point type")
    _check(shift.dtype == pd, "`input` and `shift` must have the same floating-point type")


def check_backward_dtypes(grad, x, scale, shift):
    _check(x.dtype in _DTYPE_CODE, '"lsq_backward" not implemented for \'%s\'' % str(x.dtype).replace("torch.", ""))
    pd = _param_dtype(x)
    _check(grad.dtype == x.dtype, "`grad` and `input` must have the same floating-point type")
    _check(scale.dtype == pd, "`grad` and `scale` must have the same floating-point type")
    _check(shift.dtype == pd, "`grad` and `shift` must have the same floating-point type")
    _check(x.numel() == grad.numel(), "`x` and `grad` are not the same size")


def check_channel_args(x, scale, shift, axis, backward):
    _check(scale.numel() == shift.numel(), "scale and shift need to have the same dimensions")
    # the reference forward accepts axis == x.dim() (lsq_cpu.cpp:163, off by one) and then fails in
    # x.size(axis); both directions are rejected here with the reference's message.
    _check(0 <= axis < x.dim(), "`axis` must be between 0 and number of dimensions of input")
    _check(scale.numel() == x.size(axis), "dimensions of scale and shift are not consistent with input tensor")


# -------------------------------------------------------------------------------------------------
# memory layout: the kernels see dense memory; find the [outer, C, inner] view of the channel axis
# -------------------------------------------------------------------------------------------------
def _physical_order(t):
    """dims of t from slowest to fastest varying, or None if t is not dense & non-overlapping."""
    dims = [d for d in range(t.dim()) if t.size(d) != 1]
    dims.sort(key=lambda d: (-t.stride(d), d))
    expect = 1
    for d in reversed(dims):
        if t.stride(d) != expect:
            return None
        expect *= t.size(d)
    return dims


_ROW_MAJOR = "row-major"   # marker: plain contiguous tensor, physical order == logical order


def _dense(t):
    """(tensor, physical order) with the tensor dense in memory (a contiguous copy if it was not)."""
    if t.is_contiguous():                      # the common case, one C call
        return t, _ROW_MAJOR
    order = _physical_order(t)
    if order is None:
        return t.contiguous(), _ROW_MAJOR
    return t, order


def _transposition(g, x):
    """(A, B, C) when the dense g and the dense x (same shape) hold the same elements in memory orders that differ by ONE
    swap of two adjacent groups of dimensions -- g's memory is [A][B][C], x's is [A][C][B] (contiguous NCHW against
    channels-last: A = N, B = C, C = H*W) -- else None."""
    og, ox = _physical_order(g), _physical_order(x)
    if og is None or ox is None or len(og) != len(ox) or og == ox:
        return None
    k0 = 0
    while og[k0] == ox[k0]:
        k0 += 1
    gl, xl = og[k0:], ox[k0:]
    for k in range(1, len(gl)):
        if gl[k:] + gl[:k] == xl:
            A = B = C = 1
            for d in og[:k0]:
                A *= g.size(d)
            for d in gl[:k]:
                B *= g.size(d)
            for d in gl[k:]:
                C *= g.size(d)
            return A, B, C
    return None


def _like_layout(g, x):
    """grad laid out exactly like the dense x (same strides), copying only if it is not already: GPU tensors whose orders
    differ by one transposition (a contiguous grad for a channels-last x, or the reverse) through the library's own tiled pass
    (lsq_hip_relayout: 2.5 x the rate of the generic strided copy), anything else through Tensor.copy_."""
    if g.shape == x.shape and g.stride() == x.stride():
        return g
    out = torch.empty_like(x)  # preserve_format: x is dense, so strides are kept
    if g.is_cuda and g.shape == x.shape and g.dtype == x.dtype and g.dtype in _abi._DTYPE_CODE and g.device == x.device:
        abc = _transposition(g, x)
        if abc is not None:
            rc = _on_device(x.device.index, _abi._LIB.lsq_hip_relayout, _abi._DTYPE_CODE[g.dtype], g.data_ptr(), out.data_ptr(),
                            abc[0], abc[1], abc[2], _stream_of(x.device.index))
            if rc:
                _status(rc, "lsq_hip_relayout")
            return out
    out.copy_(g.reshape(x.shape) if g.shape != x.shape else g)
    return out


def _ocl(x, order, axis):
    """[outer, C, inner] of dense x for channel `axis`, in memory order."""
    shape = x.shape
    if order is _ROW_MAJOR:
        outer = 1
        for d in shape[:axis]:
            outer *= d
        inner = 1
        for d in shape[axis + 1:]:
            inner *= d
        return outer, shape[axis], inner
    if axis not in order:  # a size-1 channel dimension: one channel covering the whole tensor
        return 1, 1, x.numel()
    k = order.index(axis)
    outer = 1
    for d in order[:k]:
        outer *= shape[d]
    inner = 1
    for d in order[k + 1:]:
        inner *= shape[d]
    return outer, shape[axis], inner


# -------------------------------------------------------------------------------------------------
# the HIP backend ("CUDA" dispatch key on ROCm).  Host cost per call matters for small layers, so the
# steady-state path is: a few C-level tensor queries, one cached lsq_params struct, two or three
# allocator calls, one ctypes call -- no context managers, no per-call struct construction.
# -------------------------------------------------------------------------------------------------
_PARAMS_CACHE = {}


def _params(qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode, numel_for_scaler=0):
    """(struct, byref) for the scalar arguments; immutable, cached per distinct argument tuple."""
    key = (qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode, numel_for_scaler)
    hit = _PARAMS_CACHE.get(key)
    if hit is not None:
        return hit
    for name, v in (("quant_min", qmin), ("quant_max", qmax), ("type_min", tmin), ("type_max", tmax)):
        _check(-2 ** 31 <= int(v) < 2 ** 31, "%s=%d does not fit a 32-bit integer" % (name, v))
    p = LsqParams(int(qmin), int(qmax), int(tmin), int(tmax), int(bool(use_gs)), int(bool(sym)),
                  int(bool(eval_mode)), int(bool(init_mode)), float(gs), int(numel_for_scaler))
    hit = (p, ctypes.byref(p))
    if len(_PARAMS_CACHE) < 4096:
        _PARAMS_CACHE[key] = hit
    return hit


def _status(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, _abi._LIB.lsq_hip_last_error().decode("utf-8", "replace")))


def _entry(name, variant):
    """(C entry point, trailing arguments): the include/lsq_hip.h symbol, or -- `variant` != 0, tools build only -- its
    `_ex` twin with the launch-variant code appended."""
    if not variant:
        return getattr(_abi._LIB, name), ()
    fn = getattr(_abi._LIB, name + "_ex", None)
    if fn is None:
        raise RuntimeError("launch variants need the tools build of the library (make -C lsqfakequantize-pytorch_amd/csrc "
                           "tools; tools/lsq_tools.py): liblsq_hip.so exports only include/lsq_hip.h")
    return fn, (int(variant),)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_of(index):
    if _raw_stream is not None:
        return _raw_stream(index)
    return torch.cuda.current_stream(index).cuda_stream


def _on_device(index, fn, *args):
    """Call the C entry point with `index` as the current HIP device (kernels launch on the current device)."""
    if torch.cuda.current_device() == index:
        return fn(*args)
    with torch.cuda.device(index):
        return fn(*args)


_WS_BYTES_PT = [0]


def _workspace(device, nbytes):
    # a fresh caching-allocator block per call: stream-ordered reuse is the allocator's job, and
    # forward/backward threads never share one.
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ---- tickets (lsq_bwd_extras): persistent per-stream arrival counters that make the backward ONE launch -------------
# The C ABI wants LSQ_TICKET_BYTES of zero-initialised device memory that outlives the call and is never shared by
# launches that can run concurrently.  One slab of _TICKET_SLOTS tickets per device is allocated (and zeroed) at the
# first eager backward on that device; streams get a slot each on first use.  Kernels of one stream are serialised by the
# stream; launches captured into a HIP graph take no ticket (see _ticket).
# Measured on MI355X: on the GPU the single-launch route is NOT faster -- the last workgroup's serial chain (drain its dx
# stores, agent-scope counter round trip, agent-scope loads of the partials) costs as much as the finalize kernel's launch
# (profiles/r02_ticket_single_launch.txt) and 2-3 us more where the kernel is busy -- but tensors of up to 8 MB are HOST-bound
# in eager mode, and there one launch less is 11-15 % of the forward + backward wall time (profiles/r03_ticket_sizes.txt:
# fp32 up to 2^21 elements, bf16 up to 2^22; +14 % at the next size up).  So the default is "auto": per-tensor backward of at
# most 8 MB (lsq_hip_policy_ticket) through a ticket, everything else through kernel + finalize.  TORCHLSQ_SINGLE_LAUNCH_BACKWARD=1 /
# set_single_launch_backward(True): always; =0 / False: never.
_TICKET_MODES = {"0": 0, "1": 1, "auto": 2}
_SINGLE_LAUNCH_BWD = [_TICKET_MODES.get(os.environ.get("TORCHLSQ_SINGLE_LAUNCH_BACKWARD", "auto").lower(), 2)]    # 0 never, 1 always, 2 auto


def set_single_launch_backward(on):
    """Tickets (one launch per backward) in both host layers from now on: True = always, False = never, "auto" = the default
    (per-tensor tensors of at most 8 MB, the host-bound ones)."""
    mode = 2 if on == "auto" else (1 if on else 0)
    _SINGLE_LAUNCH_BWD[0] = mode
    if hasattr(torch.ops, "torchlsq_native") and _abi._NATIVE_LSQ is not None:
        torch.ops.torchlsq_native._set_single_launch_backward(mode)


def _wants_ticket(nbytes, per_channel=False):
    """the decision itself is the library's (lsq_hip_policy_ticket: one rule for both host layers); the mode is this layer's state"""
    return bool(_abi._LIB.lsq_hip_policy_ticket(_SINGLE_LAUNCH_BWD[0], 1 if per_channel else 0, nbytes))


def saves_mask(eval_mode, init_mode, input_requires_grad, mask_backward):
    """does an autograd node's forward save the one-byte inside mask instead of x?  (lsq_hip_policy_saves_mask)"""
    return bool(_abi._LIB.lsq_hip_policy_saves_mask(int(bool(eval_mode)), int(bool(init_mode)), int(bool(input_requires_grad)),
                                                    int(bool(mask_backward))))


_TICKET_SLOTS = 64
_TICKET_SLABS = {}     # device index -> (slab tensor, base pointer, [next free slot])
_TICKETS = {}          # (device index, raw stream) -> byref(LsqBwdExtras)
_TICKET_KEEP = []      # the structs behind the byrefs
_TICKET_LOCK = threading.Lock()   # backward runs on autograd engine threads (one per device), forward-side callers on others


def _ticket(idx, stream):
    # A launch that is being CAPTURED into a HIP graph gets no ticket (two-launch route): the graph may later be replayed on
    # any stream, next to eager work or another replay on the capture stream, and two concurrent launches must never share
    # an arrival counter.
    # (asked about THE device the kernel is launched on: the query looks at the current device's current stream)
    if torch.cuda.current_device() == idx:
        capturing = torch.cuda.is_current_stream_capturing()
    else:
        with torch.cuda.device(idx):
            capturing = torch.cuda.is_current_stream_capturing()
    if capturing:
        return None
    key = (idx, stream)
    hit = _TICKETS.get(key)          # (a dict read is atomic under the GIL; entries are never removed or changed)
    if hit is not None:
        return hit
    with _TICKET_LOCK:
        return _ticket_locked(idx, stream, key)


def _ticket_locked(idx, stream, key):
    hit = _TICKETS.get(key)
    if hit is not None:
        return hit
    slab = _TICKET_SLABS.get(idx)
    if slab is None:
        t = torch.zeros(_TICKET_SLOTS * LSQ_TICKET_BYTES // 4, dtype=torch.int32, device=torch.device("cuda", idx))
        torch.cuda.current_stream(idx).synchronize()      # zeroed before any other stream may use a slot (one-off)
        slab = _TICKET_SLABS[idx] = (t, t.data_ptr(), [0])
    if slab[2][0] >= _TICKET_SLOTS:
        return None                                       # more streams than slots: two-launch route for the rest
    ex = LsqBwdExtras(slab[1] + slab[2][0] * LSQ_TICKET_BYTES)
    slab[2][0] += 1
    _TICKET_KEEP.append(ex)
    hit = _TICKETS[key] = ctypes.byref(ex)
    return hit


def _require_gpu(what, *tensors):
    """Every tensor of a call lives on the GPU the kernel is launched on (the first tensor's): raw pointers of
    another device would only work by accident of peer access."""
    dev = tensors[0].device
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("%s: expected a tensor on the GPU (HIP device) but got device %s" % (what, t.device))
        if t.device != dev:
            raise RuntimeError("%s: expected all tensors on %s but got one on %s" % (what, dev, t.device))


def _require_param(what, scale, shift):
    if scale.numel() < 1 or shift.numel() < 1:
        raise RuntimeError("%s: scale and shift need at least one element" % what)


def _aux_output(xd, levels_bias, want_mask):
    """(aux tensor, byref(lsq_fwd_extras)) for the optional one-byte-per-element output of the forward."""
    if levels_bias is None and not want_mask:
        return None, None
    aux = torch.empty_strided(xd.shape, xd.stride(), dtype=torch.int8, device=xd.device)
    ex = LsqFwdExtras(aux.data_ptr(), 0 if want_mask else int(levels_bias), 1 if want_mask else 0)
    return aux, ctypes.byref(ex)


def hip_forward_per_tensor(x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode,
                           levels_bias=None, variant=0, want_mask=False, levels_only=False):
    """levels_only (with levels_bias): y is not written at all (NULL in the C ABI) and only the int8 levels are returned."""
    _assert_has_ops()
    check_forward_dtypes(x, scale, shift)
    _require_gpu("lsq_forward_per_tensor", x, scale, shift)
    xd, _ = _dense(x)
    y = None if levels_only else torch.empty_like(xd)
    n = xd.numel()
    has_aux = levels_bias is not None or want_mask
    if n == 0:
        if levels_only:
            return torch.empty(x.shape, dtype=torch.int8, device=x.device)
        return (y, torch.empty(x.shape, dtype=torch.int8, device=x.device)) if has_aux else y
    _require_param("lsq_forward_per_tensor", scale, shift)
    lv, ex = _aux_output(xd, levels_bias, want_mask)
    _, pref = _params(qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode)
    scale_c, shift_c = scale.contiguous(), shift.contiguous()
    idx = x.device.index
    fn, tail = _entry("lsq_hip_forward_per_tensor", variant)
    rc = _on_device(idx, fn, _DTYPE_CODE[x.dtype], xd.data_ptr(), None if levels_only else y.data_ptr(), n,
                    scale_c.data_ptr(), shift_c.data_ptr(), pref, ex, _stream_of(idx), *tail)
    if rc:
        _status(rc, "lsq_hip_forward_per_tensor")
    if levels_only:
        return lv
    return (y, lv) if has_aux else y


def _wide_buffer(wide_out, slots, dev):
    """the caller's buffer for the un-rounded fp64 sums (the sharded path packs the element count behind them) or a new one"""
    if wide_out is None:
        return torch.empty(slots, dtype=torch.float64, device=dev)
    _check(wide_out.dtype == torch.float64 and wide_out.is_contiguous() and wide_out.numel() >= slots and wide_out.device == dev,
           "wide_out must be a contiguous float64 tensor of at least %d elements on %s" % (slots, dev))
    return wide_out


def hip_backward_per_tensor(grad, x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode,
                            numel_for_scaler=0, want_wide=False, variant=0, use_ticket=None, wide_out=None):
    _assert_has_ops()
    check_backward_dtypes(grad, x, scale, shift)
    if x.numel() <= 0:  # lsq_cpu.cpp:76-78 returns (x, scale, shift) themselves
        if want_wide:
            if wide_out is not None:
                wide_out[:2].zero_()
                return x.clone(), wide_out
            return x.clone(), torch.zeros(2, dtype=torch.float64, device=x.device)
        return x.clone(), scale.clone(), shift.clone()
    _require_gpu("lsq_backward_per_tensor", x, grad, scale, shift)
    _require_param("lsq_backward_per_tensor", scale, shift)
    xd, _ = _dense(x)
    gd = _like_layout(grad, xd)
    dx = torch.empty_like(xd)
    pd = _param_dtype(x)
    dev = x.device
    ds = torch.empty(1, dtype=pd, device=dev)
    db = torch.empty(1, dtype=pd, device=dev)
    wide = _wide_buffer(wide_out, 2, dev) if want_wide else None
    _, pref = _params(qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode, numel_for_scaler)
    code = _DTYPE_CODE[x.dtype]
    scale_c, shift_c = scale.contiguous(), shift.contiguous()
    if not _WS_BYTES_PT[0]:
        _WS_BYTES_PT[0] = int(_abi._LIB.lsq_hip_backward_per_tensor_workspace(code, xd.numel()))   # a constant
    ws = _workspace(dev, _WS_BYTES_PT[0])
    idx = dev.index
    stream = _stream_of(idx)
    if use_ticket is None:
        use_ticket = _wants_ticket(xd.numel() * xd.element_size())
    fn, tail = _entry("lsq_hip_backward_per_tensor", variant)
    rc = _on_device(idx, fn, code, gd.data_ptr(), xd.data_ptr(), dx.data_ptr(),
                    ds.data_ptr(), db.data_ptr(), wide.data_ptr() if want_wide else None, xd.numel(),
                    scale_c.data_ptr(), shift_c.data_ptr(), pref, _ticket(idx, stream) if use_ticket else None,
                    ws.data_ptr(), ws.numel(), stream, *tail)
    if rc:
        _status(rc, "lsq_hip_backward_per_tensor")
    if want_wide:
        return dx, wide
    return dx, ds, db


def hip_forward_per_channel(x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode,
                            levels_bias=None, variant=0, want_mask=False, levels_only=False):
    _assert_has_ops()
    check_forward_dtypes(x, scale, shift)
    check_channel_args(x, scale, shift, axis, backward=False)
    _require_gpu("lsq_forward_per_channel", x, scale, shift)
    xd, order = _dense(x)
    y = None if levels_only else torch.empty_like(xd)
    has_aux = levels_bias is not None or want_mask
    if x.numel() == 0:
        if levels_only:
            return torch.empty(x.shape, dtype=torch.int8, device=x.device)
        return (y, torch.empty(x.shape, dtype=torch.int8, device=x.device)) if has_aux else y
    outer, C, inner = _ocl(xd, order, axis)
    lv, ex = _aux_output(xd, levels_bias, want_mask)
    _, pref = _params(qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode)
    scale_c, shift_c = scale.contiguous(), shift.contiguous()
    idx = x.device.index
    fn, tail = _entry("lsq_hip_forward_per_channel", variant)
    rc = _on_device(idx, fn, _DTYPE_CODE[x.dtype], xd.data_ptr(), None if levels_only else y.data_ptr(), outer,
                    C, inner, scale_c.data_ptr(), shift_c.data_ptr(), pref, ex, _stream_of(idx), *tail)
    if rc:
        _status(rc, "lsq_hip_forward_per_channel")
    if levels_only:
        return lv
    return (y, lv) if has_aux else y


def hip_backward_from_mask(grad, mask):
    """dx = grad * mask: the eval-mode backward (lsq_kernel.h:126-145) from the one-byte inside mask the
    forward saved (`want_mask=True`) -- per-tensor and per-channel alike."""
    _assert_has_ops()
    _check(grad.dtype in _DTYPE_CODE, '"lsq_backward" not implemented for \'%s\'' % str(grad.dtype).replace("torch.", ""))
    _check(mask.dtype == torch.int8 and mask.numel() == grad.numel(), "`mask` must be the int8 inside mask of the forward")
    _require_gpu("lsq_backward_from_mask", grad, mask)
    if grad.shape == mask.shape and grad.stride() == mask.stride():
        gd = grad
    else:
        gd = torch.empty_strided(mask.shape, mask.stride(), dtype=grad.dtype, device=grad.device)
        gd.copy_(grad.reshape(mask.shape) if grad.shape != mask.shape else grad)
    dx = torch.empty_strided(mask.shape, mask.stride(), dtype=grad.dtype, device=grad.device)
    if grad.numel() == 0:
        return dx
    idx = grad.device.index
    rc = _on_device(idx, _abi._LIB.lsq_hip_backward_from_mask, _DTYPE_CODE[grad.dtype], gd.data_ptr(), mask.data_ptr(),
                    dx.data_ptr(), gd.numel(), _stream_of(idx))
    if rc:
        _status(rc, "lsq_hip_backward_from_mask")
    return dx


def _pc_workspace_bytes(idx, code, outer, C, inner):
    return int(_on_device(idx, _abi._LIB.lsq_hip_backward_per_channel_workspace, code, outer, C, inner))


_WS_BYTES_PC = {}


def hip_backward_per_channel(grad, x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode,
                             init_mode, numel_for_scaler=0, want_wide=False, variant=0, use_ticket=None, wide_out=None):
    _assert_has_ops()
    check_backward_dtypes(grad, x, scale, shift)
    check_channel_args(x, scale, shift, axis, backward=True)
    if x.numel() <= 0:  # lsq_cpu.cpp:221-223
        if want_wide:
            if wide_out is not None:
                wide_out[:2 * scale.numel()].zero_()
                return x.clone(), wide_out
            return x.clone(), torch.zeros(2, scale.numel(), dtype=torch.float64, device=x.device)
        return x.clone(), scale.clone(), shift.clone()
    _require_gpu("lsq_backward_per_channel", x, grad, scale, shift)
    xd, order = _dense(x)
    gd = _like_layout(grad, xd)
    dx = torch.empty_like(xd)
    outer, C, inner = _ocl(xd, order, axis)
    pd = _param_dtype(x)
    dev = x.device
    idx = dev.index
    ds = torch.empty(C, dtype=pd, device=dev)
    db = torch.empty(C, dtype=pd, device=dev)
    wide = (_wide_buffer(wide_out, 2 * C, dev) if wide_out is not None else
            torch.empty(2, C, dtype=torch.float64, device=dev)) if want_wide else None
    _, pref = _params(qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode, numel_for_scaler)
    code = _DTYPE_CODE[x.dtype]
    scale_c, shift_c = scale.contiguous(), shift.contiguous()
    wkey = (idx, code, outer, C, inner)
    nbytes = _WS_BYTES_PC.get(wkey)
    if nbytes is None:
        nbytes = _pc_workspace_bytes(idx, code, outer, C, inner)
        if len(_WS_BYTES_PC) < 4096:
            _WS_BYTES_PC[wkey] = nbytes
    ws = _workspace(dev, nbytes)
    stream = _stream_of(idx)
    if use_ticket is None:
        use_ticket = _wants_ticket(xd.numel() * xd.element_size(), per_channel=True)   # (mode "always" only: the entry point ignores it)
    fn, tail = _entry("lsq_hip_backward_per_channel", variant)
    rc = _on_device(idx, fn, code, gd.data_ptr(), xd.data_ptr(), dx.data_ptr(),
                    ds.data_ptr(), db.data_ptr(), wide.data_ptr() if want_wide else None, outer, C, inner,
                    scale_c.data_ptr(), shift_c.data_ptr(), pref, _ticket(idx, stream) if use_ticket else None,
                    ws.data_ptr(), ws.numel(), stream, *tail)
    if rc:
        _status(rc, "lsq_hip_backward_per_channel")
    if want_wide:
        return dx, wide
    return dx, ds, db


def hip_sharded_finish(packed, channels, per_channel, x_dtype, qmax, use_gs, gs):
    """(d_scale, d_shift) from the all-reduced `packed` = [sum ds terms (C), sum db terms (C), element count] of the
    batch-sharded backward: the gradient scaler from the GLOBAL count, derived on the device (lsq_hip_sharded_finish)."""
    _assert_has_ops()
    _require_gpu("lsq_sharded_finish", packed)
    _check(packed.dtype == torch.float64 and packed.is_contiguous() and packed.numel() == 2 * channels + 1,
           "lsq_sharded_finish: packed must be a contiguous float64 tensor of 2 * channels + 1 elements")
    pd = torch.float64 if x_dtype == torch.float64 else torch.float32
    dev = packed.device
    ds = torch.empty(channels, dtype=pd, device=dev)
    db = torch.empty(channels, dtype=pd, device=dev)
    _, pref = _params(0, qmax, 0, qmax, use_gs, gs, False, False, False)
    idx = dev.index
    rc = _on_device(idx, _abi._LIB.lsq_hip_sharded_finish, _DTYPE_CODE[x_dtype], packed.data_ptr(), channels,
                    1 if per_channel else 0, pref, ds.data_ptr(), db.data_ptr(), _stream_of(idx))
    if rc:
        _status(rc, "lsq_hip_sharded_finish")
    return ds, db


# ---- many per-channel quantizers in one launch (lsq_hip_*_per_channel_multi) ----------------------------------------------
_MULTI_OK = {}


def hip_multi_eligible(x, axis):
    """Can the per-channel quantizer of GPU tensor `x` along `axis` take part in a multi-tensor launch?  (Tensors the
    single-tensor policy walks with one workgroup per channel: conv / linear weights on axis 0 and the like.)"""
    if not _abi._HAS_OPS or not x.is_cuda or x.dtype not in _DTYPE_CODE or x.numel() == 0 or not (0 <= axis < x.dim()):
        return False
    if not x.is_contiguous() or (x.data_ptr() & 15):
        return False
    shape = tuple(x.shape)
    key = (x.device.index, x.dtype, shape, axis)
    hit = _MULTI_OK.get(key)
    if hit is None:
        outer, C, inner = _ocl(x, _ROW_MAJOR, axis)
        hit = bool(_on_device(x.device.index, _abi._LIB.lsq_hip_per_channel_multi_ok, _DTYPE_CODE[x.dtype], outer, C, inner, 1))
        if len(_MULTI_OK) < 65536:
            _MULTI_OK[key] = hit
    return hit


def _multi_table(n):
    return (LsqPcItem * n)()


def hip_forward_per_channel_multi(xs, scales, shifts, axes, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode):
    """y_i = fake_quant(x_i) for every tensor of the lists in ONE launch per 32 tensors.  Every x_i must satisfy
    hip_multi_eligible(x_i, axes[i]); all tensors on one GPU, one storage type."""
    _assert_has_ops()
    n = len(xs)
    dev, dt = xs[0].device, xs[0].dtype
    table = _multi_table(n)
    ys = []
    for i in range(n):
        x, scale, shift = xs[i], scales[i], shifts[i]
        check_forward_dtypes(x, scale, shift)
        check_channel_args(x, scale, shift, axes[i], backward=False)
        _require_gpu("lsq_forward_per_channel_multi", xs[0], x, scale, shift)
        _check(x.dtype == dt, "lsq_forward_per_channel_multi: all tensors must have the same floating-point type")
        _check(hip_multi_eligible(x, axes[i]), "lsq_forward_per_channel_multi: tensor %d cannot take part in a multi-tensor launch" % i)
        y = torch.empty_like(x)
        outer, C, inner = _ocl(x, _ROW_MAJOR, axes[i])
        sc, sh = scale.contiguous(), shift.contiguous()
        it = table[i]
        it.x, it.y, it.scale, it.shift = x.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr()
        it.outer, it.channels, it.inner = outer, C, inner
        ys.append((y, sc, sh))
    _, pref = _params(qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode)
    idx = dev.index
    rc = _on_device(idx, _abi._LIB.lsq_hip_forward_per_channel_multi, _DTYPE_CODE[dt], table, n, pref, _stream_of(idx))
    if rc:
        _status(rc, "lsq_hip_forward_per_channel_multi")
    return [y for y, _, _ in ys]


def hip_backward_per_channel_multi(grads, xs, scales, shifts, axes, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode,
                                   init_mode):
    """(dx_i, ds_i, db_i) for every tensor of the lists in ONE launch per 32 tensors (no workspace, no finalize launch)."""
    _assert_has_ops()
    n = len(xs)
    dev, dt = xs[0].device, xs[0].dtype
    table = _multi_table(n)
    outs, keep = [], []
    for i in range(n):
        x, g, scale, shift = xs[i], grads[i], scales[i], shifts[i]
        check_backward_dtypes(g, x, scale, shift)
        check_channel_args(x, scale, shift, axes[i], backward=True)
        _require_gpu("lsq_backward_per_channel_multi", xs[0], x, g, scale, shift)
        _check(x.dtype == dt, "lsq_backward_per_channel_multi: all tensors must have the same floating-point type")
        _check(hip_multi_eligible(x, axes[i]), "lsq_backward_per_channel_multi: tensor %d cannot take part in a multi-tensor launch" % i)
        gd = _like_layout(g, x)
        if gd.data_ptr() & 15:
            gd = gd.clone()
        dx = torch.empty_like(x)
        outer, C, inner = _ocl(x, _ROW_MAJOR, axes[i])
        pd = _param_dtype(x)
        ds = torch.empty(C, dtype=pd, device=dev)
        db = torch.empty(C, dtype=pd, device=dev)
        sc, sh = scale.contiguous(), shift.contiguous()
        it = table[i]
        it.x, it.grad, it.dx, it.scale, it.shift = x.data_ptr(), gd.data_ptr(), dx.data_ptr(), sc.data_ptr(), sh.data_ptr()
        it.ds, it.db = ds.data_ptr(), db.data_ptr()
        it.outer, it.channels, it.inner = outer, C, inner
        outs.append((dx, ds, db))
        keep.append((gd, sc, sh))
    _, pref = _params(qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode)
    idx = dev.index
    rc = _on_device(idx, _abi._LIB.lsq_hip_backward_per_channel_multi, _DTYPE_CODE[dt], table, n, pref, _stream_of(idx))
    if rc:
        _status(rc, "lsq_hip_backward_per_channel_multi")
    return outs


_WS_BYTES_MM = {}


def _two_stats(x, axis, what, ws_fn, pt_fn, pc_fn):
    """Shared host path of the one-pass statistics kernels: two results, scalar (axis None) or per channel."""
    _assert_has_ops()
    _check(x.dtype in _DTYPE_CODE, '"%s" not implemented for \'%s\'' % (what, str(x.dtype).replace("torch.", "")))
    _check(x.numel() > 0, "%s: cannot reduce an empty tensor" % what)
    _require_gpu(what, x)
    xd, order = _dense(x.detach())
    pd = _param_dtype(x)
    dev = x.device
    idx = dev.index
    code = _DTYPE_CODE[x.dtype]
    if axis is None:
        outer, C, inner = 1, 1, xd.numel()
    else:
        _check(0 <= axis < x.dim(), "`axis` must be between 0 and number of dimensions of input")
        outer, C, inner = _ocl(xd, order, axis)
    a = torch.empty(C, dtype=pd, device=dev)
    b = torch.empty(C, dtype=pd, device=dev)
    wkey = (what, idx, code, outer, C, inner)
    nbytes = _WS_BYTES_MM.get(wkey)
    if nbytes is None:
        nbytes = int(_on_device(idx, ws_fn, code, outer, C, inner))
        if len(_WS_BYTES_MM) < 4096:
            _WS_BYTES_MM[wkey] = nbytes
    ws = _workspace(dev, nbytes)
    if axis is None:
        rc = _on_device(idx, pt_fn, code, xd.data_ptr(), xd.numel(), a.data_ptr(), b.data_ptr(),
                        ws.data_ptr(), ws.numel(), _stream_of(idx))
    else:
        rc = _on_device(idx, pc_fn, code, xd.data_ptr(), outer, C, inner, a.data_ptr(),
                        b.data_ptr(), ws.data_ptr(), ws.numel(), _stream_of(idx))
    if rc:
        _status(rc, what)
    if axis is None:
        return a.reshape(()), b.reshape(())
    return a, b


_OBS_UPDATE_CACHE = {}


def hip_observer_update(cur_min, cur_max, min_state, max_state, scale_out, shift_out, mode, first, averaging_constant,
                        quant_min, quant_max, symmetric, zero_point_symmetric, eps):
    """One launch: fold the batch's min / max into an observer's running state (in place), derive torch's qparams
    from it and store the LSQ parameters scale / shift = -zero_point * scale (lsq_hip_observer_update).  All fp32,
    one element per channel; nothing is read back to the host."""
    _assert_has_ops()
    tensors = (cur_min, cur_max, min_state, max_state, scale_out, shift_out)
    _require_gpu("lsq_observer_update", *tensors)
    n = min_state.numel()
    for t in tensors:
        _check(t.dtype == torch.float32 and t.numel() == n and t.is_contiguous(),
               "lsq_observer_update: every tensor must be a contiguous float32 tensor with one element per channel")
    key = (mode, first, float(averaging_constant), quant_min, quant_max, bool(symmetric), zero_point_symmetric, float(eps))
    hit = _OBS_UPDATE_CACHE.get(key)
    if hit is None:
        u = LsqObserverUpdate(int(mode), int(first), float(averaging_constant), int(quant_min), int(quant_max),
                              int(bool(symmetric)), int(zero_point_symmetric), float(eps))
        hit = _OBS_UPDATE_CACHE[key] = (u, ctypes.byref(u))
    idx = min_state.device.index
    rc = _on_device(idx, _abi._LIB.lsq_hip_observer_update, n, cur_min.data_ptr(), cur_max.data_ptr(), min_state.data_ptr(),
                    max_state.data_ptr(), hit[1], scale_out.data_ptr(), shift_out.data_ptr(), _stream_of(idx))
    if rc:
        _status(rc, "lsq_hip_observer_update")


def hip_minmax(x, axis=None):
    """(min, max) of x -- over everything (axis None) or per channel along `axis` -- in one read-only pass.
    torch.aminmax semantics (a NaN makes both results NaN); results are fp32 (fp64 for fp64 input)."""
    _assert_has_ops()
    return _two_stats(x, axis, "lsq_minmax", _abi._LIB.lsq_hip_minmax_workspace, _abi._LIB.lsq_hip_minmax_per_tensor,
                      _abi._LIB.lsq_hip_minmax_per_channel)


def hip_meanstd(x, axis=None):
    """(mean, unbiased std) of x -- over everything (axis None) or per channel along `axis`, over the other axes --
    in one read-only pass (reference observers.py:329-337 uses torch.mean + torch.std); fp64 accumulation,
    results fp32 (fp64 for fp64 input)."""
    _assert_has_ops()
    return _two_stats(x, axis, "lsq_meanstd", _abi._LIB.lsq_hip_meanstd_workspace, _abi._LIB.lsq_hip_meanstd_per_tensor,
                      _abi._LIB.lsq_hip_meanstd_per_channel)



# ---- which launch a per-channel backward gets (lsq_hip_plan_backward_per_channel) --------------------------------------------
_PLAN_KINDS = {0: "none", 1: "windows", 2: "row-groups", 3: "segment", 4: "owners"}


def hip_plan_backward_per_channel(x, axis, sym=False, eval_mode=False, init_mode=False, aligned=None):
    """The launch the library's policy gives the per-channel backward of GPU tensor `x` along `axis` -- kernel family, grid,
    workgroup size, loop form -- as a dict; nothing is launched.  The SHIPPED library answers (no debug build needed)."""
    _assert_has_ops()
    _require_gpu("lsq_plan_backward_per_channel", x)
    xd, order = _dense(x)
    outer, C, inner = _ocl(xd, order, axis)
    _, pref = _params(0, 127, 0, 255, True, 1.0, sym, eval_mode, init_mode)
    out = (ctypes.c_int32 * 8)()
    al = (xd.data_ptr() & 15) == 0 if aligned is None else bool(aligned)
    rc = _on_device(x.device.index, _abi._LIB.lsq_hip_plan_backward_per_channel, _DTYPE_CODE[x.dtype], outer, C, inner,
                    1 if al else 0, pref, ctypes.byref(out))
    if rc:
        _status(rc, "lsq_hip_plan_backward_per_channel")
    return dict(grid_x=out[0], grid_y=out[1], resident_per_cu=out[2], vgprs=out[3], kind=_PLAN_KINDS.get(out[4], "?"),
                ring_depth=out[5], block=out[6], ring_nt=out[7])


# ---- rank communicator (lsq_hip_comm_*): the one collective of the batch-sharded backward, issued by the library ----------
LSQ_COMM_SUM, LSQ_COMM_MIN, LSQ_COMM_MAX = 0, 1, 2
LSQ_COMM_ID_BYTES = 128
_COMM_DTYPES = {torch.float32: _abi.LSQ_F32, torch.float64: _abi.LSQ_F64}


class HipComm:
    """A communicator of the library's own over RCCL (include/lsq_hip.h, "rank communicator") for one rank on one GPU.
    `unique_id()` on one rank, the 128 bytes to every rank by any means, `HipComm(id, rank, nranks, device)` on all of them
    (a collective call: it returns once every rank has joined).  all_reduce runs on the current stream of the device;
    begin / end run it on the communicator's own stream, ordered behind the work already enqueued on the current stream."""

    @staticmethod
    def unique_id():
        _assert_has_ops()
        buf = (ctypes.c_ubyte * LSQ_COMM_ID_BYTES)()
        rc = _abi._LIB.lsq_hip_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p))
        if rc:
            _status(rc, "lsq_hip_comm_unique_id")
        return bytes(buf)

    def __init__(self, uid, rank, nranks, device, event_system_fence=False):
        _assert_has_ops()
        _check(len(uid) == LSQ_COMM_ID_BYTES, "HipComm: the id is %d bytes" % LSQ_COMM_ID_BYTES)
        self.device = torch.device(device)
        self.index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.rank, self.nranks = int(rank), int(nranks)
        handle = ctypes.c_void_p()
        buf = (ctypes.c_ubyte * LSQ_COMM_ID_BYTES).from_buffer_copy(uid)
        rc = _on_device(self.index, _abi._LIB.lsq_hip_comm_create, ctypes.cast(buf, ctypes.c_void_p), self.rank, self.nranks,
                        ctypes.byref(self._options(event_system_fence)), ctypes.byref(handle))
        if rc:
            _status(rc, "lsq_hip_comm_create")
        self.handle = handle.value
        self.checked = None          # what torchlsq.distributed.native_comm found out about it (a dict), for records

    @staticmethod
    def _options(event_system_fence):
        return _abi.LsqCommOptions(size=ctypes.sizeof(_abi.LsqCommOptions), event_system_fence=int(bool(event_system_fence)))

    def configure(self, event_system_fence):
        """SETUP (waits for the communicator's stream): events with / without the system-scope fence (include/lsq_hip.h,
        lsq_comm_options)"""
        rc = _on_device(self.index, _abi._LIB.lsq_hip_comm_configure, self.handle, ctypes.byref(self._options(event_system_fence)))
        if rc:
            _status(rc, "lsq_hip_comm_configure")

    def tune(self):
        """SETUP (allocates scratch, synchronises): pick the communicator's stream by measurement against the device's CURRENT
        stream -- call it once, on the stream the steps will run on, at the same point on every rank"""
        rc = _on_device(self.index, _abi._LIB.lsq_hip_comm_tune, self.handle, _stream_of(self.index))
        if rc:
            _status(rc, "lsq_hip_comm_tune")

    def side_stream(self):
        """the communicator's own stream as a torch stream (for consumers of a begun reduction; join with end() once)"""
        ptr = int(_abi._LIB.lsq_hip_comm_side_stream(self.handle))     # (lsq_hip_comm_tune may change it: ask every time)
        hit = getattr(self, "_side", None)
        if hit is None or hit[0] != ptr:
            hit = self._side = (ptr, torch.cuda.ExternalStream(ptr, device=self.device))
        return hit[1]

    def info(self):
        out = (ctypes.c_int32 * 8)()
        rc = _abi._LIB.lsq_hip_comm_info(self.handle, ctypes.byref(out))
        if rc:
            _status(rc, "lsq_hip_comm_info")
        d = dict(rank=out[0], nranks=out[1], device=out[2], side_stream_choice=out[3], event_system_fence=out[4], rccl_version=out[5],
                 reductions_begun=out[6])
        if self.checked is not None:
            d["checked"] = dict(self.checked)
        return d

    def _args(self, t, out, op):
        _check(t.is_cuda and t.device.index == self.index and t.is_contiguous() and t.dtype in _COMM_DTYPES,
               "HipComm: a contiguous float32 / float64 tensor on cuda:%d" % self.index)
        recv = t if out is None else out
        _check(recv.dtype == t.dtype and recv.numel() == t.numel() and recv.is_contiguous() and recv.device == t.device,
               "HipComm: `out` must match the input")
        return t.data_ptr(), recv.data_ptr(), t.numel(), _COMM_DTYPES[t.dtype], int(op)

    def all_reduce(self, t, op=LSQ_COMM_SUM, out=None):
        """in place (or into `out`) on the device's current stream"""
        a = self._args(t, out, op)
        rc = _on_device(self.index, _abi._LIB.lsq_hip_comm_all_reduce, self.handle, *a, _stream_of(self.index))
        if rc:
            _status(rc, "lsq_hip_comm_all_reduce")
        return t if out is None else out

    def begin(self, t, op=LSQ_COMM_SUM, out=None):
        """start the reduction on the communicator's stream behind the current stream's work; returns a ticket for end()"""
        a = self._args(t, out, op)
        ticket = ctypes.c_int32(-1)
        rc = _on_device(self.index, _abi._LIB.lsq_hip_comm_all_reduce_begin, self.handle, *a, _stream_of(self.index),
                        ctypes.byref(ticket))
        if rc:
            _status(rc, "lsq_hip_comm_all_reduce_begin")
        return ticket.value

    def end(self, ticket):
        """the device's current stream waits for reduction `ticket` (the host does not)"""
        rc = _on_device(self.index, _abi._LIB.lsq_hip_comm_all_reduce_end, self.handle, int(ticket), _stream_of(self.index))
        if rc:
            _status(rc, "lsq_hip_comm_all_reduce_end")

    def join(self):
        """the device's current stream waits for everything enqueued on the communicator's stream so far (reductions begun,
        consumers placed behind them): the once-per-step join"""
        rc = _on_device(self.index, _abi._LIB.lsq_hip_comm_join, self.handle, _stream_of(self.index))
        if rc:
            _status(rc, "lsq_hip_comm_join")

    def destroy(self):
        if self.handle:
            h, self.handle = self.handle, None
            rc = _on_device(self.index, _abi._LIB.lsq_hip_comm_destroy, h)
            if rc:
                _status(rc, "lsq_hip_comm_destroy")
