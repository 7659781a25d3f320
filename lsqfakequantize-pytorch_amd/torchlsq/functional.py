"""torchlsq.functional -- the functional entry point of the LSQ / LSQ+ fake quantizer.

Drop-in for reference torchlsq/functional.py:8-97: same name, same argument list, defaults and
assertions; the work is done by `torch.ops.torchlsq.lsq` (registered in torchlsq/extension.py on top
of the gfx950 kernels).
"""
import torch
from torch.autograd.function import once_differentiable

from . import extension as _E
from .extension import _assert_has_ops

Tensor = torch.Tensor


class _LSQOnDevice(torch.autograd.Function):
    """Direct autograd binding of the gfx950 kernels for GPU tensors.

    Same computation as `torch.ops.torchlsq.lsq` (the registered ops stay available and are what the
    dispatcher-level tests exercise); this class only skips the dispatcher round trips -- two Python
    re-entries per op -- which dominate the cost of small layers.  Mirrors LSQPer*Function of the
    reference (csrc/ops/autograd/lsq_autograd.cpp:16-74,111-173): saves {input, scale, shift}, backward
    returns gradients for those three only, double backward is refused.
    """

    @staticmethod
    def forward(ctx, x, scale, shift, cfg):
        (qmin, qmax, tmin, tmax, axis, use_gs, gs, sym, per_channel, eval_mode, init_mode, mask_backward) = cfg
        # eval mode (plain fake-quantizer behaviour): the backward only needs "was the element strictly inside the
        # range", so the forward emits that as one byte per element and autograd keeps the mask instead of x (1 instead
        # of 4 bytes per fp32 element, and the backward reads 9 instead of 12 bytes per element).
        # The mask is that of the FORWARD's parameter values, whereas the reference recomputes it in its backward from
        # the saved (x, scale, shift) (lsq_autograd.cpp:46-73), i.e. from the parameter values AT BACKWARD TIME.  The two
        # differ only if scale / shift are overwritten in place (param.data.copy_, which autograd's version check does
        # not see) between this forward and its backward -- which is exactly what the observer-driven phase of
        # LSQFakeQuantizer does on every call (observers.py:417-420).  `mask_backward=False` (what the module passes
        # while its observer is enabled) keeps the reference's behaviour: x is saved and the eval backward runs on the
        # current parameters.  (The C++ host binding's LsqNode has the same switch.)
        masked = _E.saves_mask(eval_mode, init_mode, x.requires_grad, mask_backward)
        if per_channel:
            y = _E.hip_forward_per_channel(x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode,
                                           init_mode, want_mask=masked)
        else:
            y = _E.hip_forward_per_tensor(x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode,
                                          want_mask=masked)
        ctx.masked = masked
        if masked:
            y, mask = y
            ctx.save_for_backward(mask, scale, shift)
        else:
            ctx.save_for_backward(x, scale, shift)
        ctx.cfg = cfg
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, scale, shift = ctx.saved_tensors
        (qmin, qmax, tmin, tmax, axis, use_gs, gs, sym, per_channel, eval_mode, init_mode, _) = ctx.cfg
        if ctx.masked:      # x is the inside mask here; d_scale = d_shift = 0 (lsq_kernel.h:142-144)
            return _E.hip_backward_from_mask(grad_out, x), torch.zeros_like(scale), torch.zeros_like(shift), None
        if per_channel:
            dx, ds, db = _E.hip_backward_per_channel(grad_out, x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs,
                                                     sym, eval_mode, init_mode)
        else:
            dx, ds, db = _E.hip_backward_per_tensor(grad_out, x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym,
                                                    eval_mode, init_mode)
        return dx, ds, db, None


class _LSQOnHost(torch.autograd.Function):
    """The same for tensors in host memory (liblsq_cpu.so, the counterpart of the reference's CPU dispatch key): straight to
    the kernels' host layer instead of through the dispatcher -- torch.library's Python autograd wrapper costs ~70 us per call
    (schema defaults, two re-entries), several times the kernel on a small tensor.  Mirrors lsq_autograd.cpp:16-74,111-173:
    saves {input, scale, shift}; the eval backward recomputes the mask from them (no mask variant here)."""

    @staticmethod
    def forward(ctx, x, scale, shift, cfg):
        (qmin, qmax, tmin, tmax, axis, use_gs, gs, sym, per_channel, eval_mode, init_mode, _) = cfg
        ctx.save_for_backward(x, scale, shift)
        ctx.cfg = cfg
        return _E.cpu_forward(x, scale, shift, axis, per_channel, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, scale, shift = ctx.saved_tensors
        (qmin, qmax, tmin, tmax, axis, use_gs, gs, sym, per_channel, eval_mode, init_mode, _) = ctx.cfg
        dx, ds, db = _E.cpu_backward(grad_out, x, scale, shift, axis, per_channel, qmin, qmax, tmin, tmax, use_gs, gs, sym,
                                     eval_mode, init_mode)
        return dx, ds, db, None


def lsq(x: Tensor, scale: Tensor, shift: Tensor,
        quant_min: int = 0,
        quant_max: int = 255,
        type_min: int = None,
        type_max: int = None,
        axis: int = 1,
        use_grad_scaling: bool = True,
        grad_scaler: float = 1.,
        is_affine: bool = True,
        is_perchannel: bool = False,
        eval_mode: bool = False,
        init_mode: bool = False, *,
        mask_backward: bool = True) -> Tensor:
    """Learned Step Size Quantization (LSQ+, arXiv:2004.09576) fake quantizer: quantize -> dequantize
    with `scale` and `shift` as learnable parameters.

    Forward (per element, reference csrc/ops/kernels/lsq_kernel.h:6-14)::

        s   = max(|scale|, eps)
        zp  = round(clamp(-shift / s, type_min, type_max))          # the integer zero point
        x_q = round(clamp(x / s + zp, quant_min, quant_max))        # round half to even
        x_r = (x_q - zp) * s

    Backward (lsq_kernel.h:94-123), with xq the *unrounded* clamped value::

        d x     = grad                       if quant_min < xq < quant_max else 0
        d scale = grad * (x_r - x) / s       inside;   grad * (quant_min|quant_max - zp) at the borders
        d shift = 0 inside (or if symmetric); grad at the borders

    `d scale` and `d shift` are summed over the tensor (per channel along `axis` when
    `is_perchannel`) and, when `use_grad_scaling`, multiplied by
    grad_scaler / sqrt(numel * quant_max [/ channels]) (arXiv:1902.08153).

    Args:
        x: input tensor (float32/float64; this build also takes bfloat16/float16 with fp32 parameters).
        scale, shift: 1-D tensors -- one element for per-tensor, one per channel otherwise
            (a single-element tensor is broadcast in the per-channel case).
        quant_min, quant_max: bounds of the quantized range (default 0..255).
        type_min, type_max: numeric limits of the quantized *type* used to clamp the zero point;
            default to quant_min / quant_max.
        axis: channel dimension of the per-channel scheme.
        use_grad_scaling, grad_scaler: gradient scaling of the parameters, see above.
        is_affine: asymmetric (True) or symmetric (False: no gradient for `shift`) quantization.
        is_perchannel: per-channel (True) or per-tensor (False).
        eval_mode: behave like a plain fake-quantizer (no parameter gradients).
        init_mode: parameter-initialisation phase: the forward is the identity and the parameter
            gradients are those of ||x_r - x||^2 (the upstream gradient is ignored for them).
        mask_backward (keyword only; this build, GPU tensors): in eval mode keep the forward's one-byte "inside the
            range" mask for the backward (default) instead of x.  False = the reference's behaviour to the letter: x is
            saved and the backward recomputes the mask from the parameters as they are THEN (lsq_autograd.cpp:46-73) --
            it only matters when scale / shift are overwritten in place between a forward and its backward.
    """
    _assert_has_ops()
    if not is_affine:
        assert quant_min <= 0 <= quant_max, 'quantization range must be covered 0 in symmetric quantization'
    if type_min is None:
        type_min = quant_min
    if type_max is None:
        type_max = quant_max
    on_gpu = x.is_cuda and scale.is_cuda and shift.is_cuda
    on_host = not (x.is_cuda or scale.is_cuda or shift.is_cuda) and x.device.type == "cpu"
    if (on_gpu or on_host) and not torch.jit.is_tracing() and not torch.compiler.is_compiling():
        # front-op checks and routing of quantops::ops::lsq (lsq.cpp:104-134), then straight to the kernels
        native = _E._NATIVE_LSQ if on_gpu else None
        if native is not None:      # C++ front op + autograd node (csrc/torch_binding): same kernels, less host time
            if not mask_backward:
                native = torch.ops.torchlsq_native.lsq_keep_input.default
            return native(x, scale, shift, quant_min, quant_max, type_min, type_max, axis, bool(use_grad_scaling),
                          float(grad_scaler), bool(is_affine), bool(is_perchannel), bool(eval_mode), bool(init_mode))
        if scale.dim() != 1:
            raise RuntimeError("scale should be a 1-D tensor, even in per tensor case(please, avoid torch.Scalar too)")
        if shift.dim() != 1:
            raise RuntimeError("shift should be a 1-D tensor, even in per tensor case(please, avoid torch.Scalar too)")
        if is_perchannel:
            size = max(scale.size(0), shift.size(0))
            if scale.size(0) != size:
                scale = scale.repeat(size)      # differentiable: a size-1 leaf receives the summed gradient
            if shift.size(0) != size:
                shift = shift.repeat(size)
        cfg = (quant_min, quant_max, type_min, type_max, axis, bool(use_grad_scaling), float(grad_scaler),
               not is_affine, bool(is_perchannel), bool(eval_mode), bool(init_mode), bool(mask_backward))
        return (_LSQOnDevice if on_gpu else _LSQOnHost).apply(x, scale, shift, cfg)
    return torch.ops.torchlsq.lsq(x, scale, shift, quant_min, quant_max, type_min, type_max,
                                  axis, use_grad_scaling, grad_scaler, is_affine, is_perchannel,
                                  eval_mode, init_mode)


def lsq_quantize(x: Tensor, scale: Tensor, shift: Tensor,
                 quant_min: int = 0,
                 quant_max: int = 255,
                 type_min: int = None,
                 type_max: int = None,
                 axis: int = 1,
                 is_perchannel: bool = False,
                 dtype=torch.quint8) -> Tensor:
    """The REAL quantized tensor behind `lsq`'s fake-quantized output (an addition of this build): a `torch.quint8` /
    `torch.qint8` tensor whose integer representation holds the levels x_q of the forward (lsq_kernel.h:13) and whose
    quantizer carries s = max(|scale|, eps) and the integer zero point zp = round(clamp(-shift / s, type_min, type_max)) the
    kernels use -- so for FLOAT32 x `lsq_quantize(...).dequantize()` equals `lsq(...)` bit for bit ((x_q - zp) * s, the same
    two fp32 operations; torch's quantizer dequantizes in fp32, so a float64 x agrees only to fp32 precision and a 16-bit x
    gets fp32 where `lsq` returns the value rounded to 16 bits -- the integer levels are the forward's in every case).  The
    per-tensor form reads scale / zero point back to the host once per call (torch's per-tensor quantizer holds host numbers):
    a conversion-time function, not one for a training loop.  One pass that reads x and writes ONE byte per element (no
    fake-quantized output: 5 instead of 9 bytes per fp32 element on the GPU).  The step after the path: reference quantized/modules/observers.py:378-422 hands scale and
    zero_point to torch's converter, which quantizes again with its own rounding; here the trained quantizer emits them.
    """
    _assert_has_ops()
    assert dtype in (torch.quint8, torch.qint8), "dtype must be torch.quint8 or torch.qint8"
    type_min = quant_min if type_min is None else type_min
    type_max = quant_max if type_max is None else type_max
    lo, hi = (0, 255) if dtype == torch.quint8 else (-128, 127)
    assert lo <= quant_min <= quant_max <= hi, "the quantized range must fit the quantized type"
    if scale.dim() != 1 or shift.dim() != 1:
        raise RuntimeError("scale and shift should be 1-D tensors, even in per tensor case(please, avoid torch.Scalar too)")
    x, scale, shift = x.detach(), scale.detach(), shift.detach()
    if is_perchannel:
        size = max(scale.size(0), shift.size(0))
        scale = scale if scale.size(0) == size else scale.repeat(size)
        shift = shift if shift.size(0) == size else shift.repeat(size)
        levels = torch.ops.torchlsq.lsq_levels_per_channel(x, scale, shift, axis, quant_min, quant_max, type_min, type_max, 0)
    else:
        levels = torch.ops.torchlsq.lsq_levels_per_tensor(x, scale, shift, quant_min, quant_max, type_min, type_max, 0)
    # the quantizer's constants exactly as the kernels derive them (lsq_cpu.cpp:44-47 / lsq_kernel.h:157-158,12): |scale|
    # floored at eps, the zero point from -shift * (1 / s), clamped to the type's range, rounded half to even
    s = scale.abs().clamp_min(torch.finfo(scale.dtype).eps)
    zp = torch.fmin(torch.full_like(s, type_max), torch.fmax(torch.full_like(s, type_min), -shift * (1.0 / s))).round()
    int_repr = levels.view(torch.uint8) if dtype == torch.quint8 else levels        # the byte is q mod 256 either way
    if is_perchannel:
        return torch._make_per_channel_quantized_tensor(int_repr, s.to(torch.float64), zp.to(torch.int64), axis)
    s0, zp0 = torch.stack([s[0].double(), zp[0].double()]).tolist()          # the quantizer wants host numbers: ONE read-back
    return torch._make_per_tensor_quantized_tensor(int_repr, s0, int(zp0))


class _LSQForeach(torch.autograd.Function):
    """N per-channel quantizers as ONE autograd node over the multi-tensor kernels (lsq_hip_*_per_channel_multi): one launch
    per 32 tensors each way instead of N.  Same arithmetic and summation order as N `lsq` calls (bit-identical outputs and
    gradients); saves {x_i, scale_i, shift_i} like the reference's nodes (lsq_autograd.cpp:111-173)."""

    @staticmethod
    def forward(ctx, cfg, n, *tensors):
        xs, scales, shifts = tensors[:n], tensors[n:2 * n], tensors[2 * n:]
        (qmin, qmax, tmin, tmax, axes, use_gs, gs, sym, eval_mode, init_mode) = cfg
        ys = _E.hip_forward_per_channel_multi(xs, scales, shifts, axes, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode,
                                              init_mode)
        ctx.save_for_backward(*tensors)
        ctx.cfg, ctx.n = cfg, n
        ctx.set_materialize_grads(False)        # an unused output arrives as None in backward, not as a zero tensor
        return tuple(ys)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grad_outs):
        n = ctx.n
        tensors = ctx.saved_tensors
        xs, scales, shifts = tensors[:n], tensors[n:2 * n], tensors[2 * n:]
        (qmin, qmax, tmin, tmax, axes, use_gs, gs, sym, eval_mode, init_mode) = ctx.cfg
        # an output nobody used has no gradient: its tensor takes no part in the launch and gets NO gradients, exactly what N
        # separate lsq calls give it (with init_mode the parameter gradients ignore the upstream gradient, lsq_kernel.h:116: a
        # zero-filled stand-in would invent d_scale / d_shift for it)
        live = [i for i in range(n) if grad_outs[i] is not None]
        dxs, dss, dbs = [None] * n, [None] * n, [None] * n
        if live:
            outs = _E.hip_backward_per_channel_multi([grad_outs[i] for i in live], [xs[i] for i in live], [scales[i] for i in live],
                                                     [shifts[i] for i in live], [axes[i] for i in live], qmin, qmax, tmin, tmax,
                                                     use_gs, gs, sym, eval_mode, init_mode)
            for i, o in zip(live, outs):
                dxs[i], dss[i], dbs[i] = o
        return (None, None) + tuple(dxs) + tuple(dss) + tuple(dbs)


def lsq_foreach(xs, scales, shifts,
                quant_min: int = 0,
                quant_max: int = 255,
                type_min: int = None,
                type_max: int = None,
                axis=0,
                use_grad_scaling: bool = True,
                grad_scaler: float = 1.,
                is_affine: bool = True,
                eval_mode: bool = False,
                init_mode: bool = False):
    """`lsq(x_i, scale_i, shift_i, ..., is_perchannel=True)` for every i, horizontally fused (an addition of this build).

    The per-channel quantizers of many tensors -- typically all conv / linear weights of a QAT model, each a few MB and
    launch-latency-bound on its own -- run in ONE launch per 32 tensors each way.  `axis` is one int or one per tensor; all
    other arguments are shared and mean what they mean in `lsq`.  Returns the list of outputs; gradients reach every x_i,
    scale_i, shift_i exactly as through N separate `lsq` calls (bit-identical).  Tensors the multi-tensor kernels do not take
    (CPU tensors, channel rows too short or unaligned, tensors so small or so large that the single-tensor policy splits their
    channels differently: `torchlsq.extension.hip_multi_eligible`) silently go through `lsq` one by one.
    """
    _assert_has_ops()
    n = len(xs)
    assert len(scales) == n and len(shifts) == n, "xs, scales and shifts must have the same length"
    if not is_affine:
        assert quant_min <= 0 <= quant_max, 'quantization range must be covered 0 in symmetric quantization'
    type_min = quant_min if type_min is None else type_min
    type_max = quant_max if type_max is None else type_max
    axes = [int(axis)] * n if isinstance(axis, int) else [int(a) for a in axis]
    assert len(axes) == n
    fusable = not torch.jit.is_tracing() and not torch.compiler.is_compiling()
    all_gpu = all(x.is_cuda and sc.is_cuda and sh.is_cuda for x, sc, sh in zip(xs, scales, shifts))
    if fusable and n > 1 and _E._NATIVE_LSQ is not None and all_gpu:     # (a list with CPU entries: the Python partition below)
        # C++ host layer: the partition into fused / single tensors, the checks and the one autograd node all happen there
        # (a table row and an output allocation of host time per tensor instead of a Python call chain)
        return list(torch.ops.torchlsq_native.lsq_foreach(list(xs), list(scales), list(shifts), axes, quant_min, quant_max,
                                                          type_min, type_max, bool(use_grad_scaling), float(grad_scaler),
                                                          bool(is_affine), bool(eval_mode), bool(init_mode)))
    out = [None] * n
    groups = {}
    for i in range(n):
        x, sc, sh = xs[i], scales[i], shifts[i]
        if (fusable and x.is_cuda and sc.is_cuda and sh.is_cuda and sc.dim() == 1 and sh.dim() == 1
                and _E.hip_multi_eligible(x, axes[i])):
            groups.setdefault((x.device, x.dtype), []).append(i)
        else:
            out[i] = lsq(x, sc, sh, quant_min, quant_max, type_min, type_max, axes[i], use_grad_scaling, grad_scaler, is_affine,
                         True, eval_mode, init_mode)
    for idx in groups.values():
        if len(idx) == 1:       # nothing to fuse
            i = idx[0]
            out[i] = lsq(xs[i], scales[i], shifts[i], quant_min, quant_max, type_min, type_max, axes[i], use_grad_scaling,
                         grad_scaler, is_affine, True, eval_mode, init_mode)
            continue
        sc_l, sh_l = [], []
        for i in idx:       # front-op rule (lsq.cpp:124-126): a size-1 parameter is repeated up to the other's size
            sc, sh = scales[i], shifts[i]
            size = max(sc.size(0), sh.size(0))
            sc_l.append(sc if sc.size(0) == size else sc.repeat(size))
            sh_l.append(sh if sh.size(0) == size else sh.repeat(size))
        cfg = (quant_min, quant_max, type_min, type_max, tuple(axes[i] for i in idx), bool(use_grad_scaling), float(grad_scaler),
               not is_affine, bool(eval_mode), bool(init_mode))
        ys = _LSQForeach.apply(cfg, len(idx), *[xs[i] for i in idx], *sc_l, *sh_l)
        for i, y in zip(idx, ys):
            out[i] = y
    return out
