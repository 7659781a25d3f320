"""Op loader and registration for the MI355X build of torchlsq.

This module replaces two pieces of the reference:
  * torchlsq/extension.py (reference :12-56), which located `_C.so` and `torch.ops.load_library`-ed
    it: here the native part is `liblsq_hip.so`, a C-ABI HIP library (include/lsq_hip.h) opened
    with ctypes;
  * the registration blocks of the C++ extension -- the schemas (csrc/ops/lsq.cpp:137-146,
    csrc/torchlsq.cpp:35-39), the composite front op `lsq` (lsq.cpp:104-134), the autograd-key
    kernels (csrc/ops/autograd/lsq_autograd.cpp) and the backend kernels' argument checks
    (lsq_cpu.cpp:28-29,72-78,159-163,214-223) -- which are restated with `torch.library`.

Dispatch keys: HIP tensors carry PyTorch's "CUDA" dispatch key on ROCm builds, so the gfx950
kernels are registered under "CUDA".  CPU tensors are served, like in the reference
(TORCH_LIBRARY_IMPL(torchlsq, CPU), lsq_cpu.cpp:298-311), by kernels for host memory: `liblsq_cpu.so`
(include/lsq_cpu.h, csrc/cpu/lsq_cpu_twin.cpp), registered under "CPU".  The two never substitute for
each other: a GPU tensor is only ever handled by the HIP library, and when `liblsq_hip.so` is missing
the package refuses to work at all (`_assert_has_ops`), CPU tensors included.
"""
import torch

from . import _abi
from ._abi import *  # noqa: F401,F403  (structures, C_ABI tables, loaders: the names tests and tools import from here)
from ._abi import (ABI_VERSION, C_ABI, C_ABI_CPU, _assert_has_ops, _check_hip_version, _has_ops, host_binding,  # noqa: F401
                   library, native_lsq, set_host_binding, set_library)
from ._hip_host import *  # noqa: F401,F403
from ._hip_host import (_SINGLE_LAUNCH_BWD, _TICKET_SLABS, _TICKETS, _WS_BYTES_PC, _WS_BYTES_PT, _check, _dense,  # noqa: F401
                        _wants_ticket,
                        _like_layout, _ocl, _param_dtype, _params, _physical_order, _ROW_MAJOR, _transposition, hip_backward_from_mask,
                        hip_backward_per_channel, hip_backward_per_channel_multi, hip_backward_per_tensor,
                        hip_forward_per_channel, hip_forward_per_channel_multi, hip_forward_per_tensor, hip_meanstd,
                        hip_minmax, hip_multi_eligible, hip_observer_update, hip_plan_backward_per_channel, hip_sharded_finish, HipComm, LSQ_COMM_ID_BYTES,
                        LSQ_COMM_MAX, LSQ_COMM_MIN, LSQ_COMM_SUM,
                        saves_mask, set_single_launch_backward)
from ._cpu_host import _cpu_meanstd, _cpu_minmax, cpu_backward, cpu_forward, cpu_levels, cpu_sharded_finish  # noqa: F401


def __getattr__(name):
    # loader state lives in _abi (it changes at run time: set_host_binding, set_library); read it through this module too
    if name in ("_LIB", "_HAS_OPS", "_CPU_LIB", "_NATIVE_LSQ", "error_str", "cpu_error_str", "native_error_str"):
        return getattr(_abi, name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


# -------------------------------------------------------------------------------------------------
# schemas -- namespace and signatures of the reference (lsq.cpp:138-145, torchlsq.cpp:36-37)
# -------------------------------------------------------------------------------------------------
_TAIL = ("int quant_min, int quant_max, int type_min, int type_max, bool use_grad_scaling, float grad_scaler, "
         "bool sym, bool eval_mode, bool init_mode")
_lib_def = torch.library.Library("torchlsq", "DEF")
_lib_def.define("_cuda_version() -> int")
_lib_def.define("lsq(Tensor x, Tensor scale, Tensor shift, int quant_min, int quant_max, int type_min, int type_max, "
                "int axis, bool use_grad_scaling, float grad_scale, bool is_affine, bool is_perchannel, "
                "bool eval_mode, bool init_mode) -> Tensor")
_lib_def.define("lsq_forward_per_tensor(Tensor x, Tensor scale, Tensor shift, " + _TAIL + ") -> Tensor")
_lib_def.define("lsq_backward_per_tensor(Tensor grad, Tensor x, Tensor scale, Tensor shift, " + _TAIL +
                ") -> (Tensor, Tensor, Tensor)")
_lib_def.define("lsq_forward_per_channel(Tensor x, Tensor scale, Tensor shift, int axis, " + _TAIL + ") -> Tensor")
_lib_def.define("lsq_backward_per_channel(Tensor grad, Tensor x, Tensor scale, Tensor shift, int axis, " + _TAIL +
                ") -> (Tensor, Tensor, Tensor)")
# Additions of this build (not in the reference):
#  * `*_wide`: backward that also takes the element count for the gradient scaler and returns the
#    un-rounded fp64 reductions ([2] or [2, C]) -- what the batch-sharded path all-reduces;
#  * `lsq_quantize_*`: forward that also emits the int8 integer levels (q - level_bias).
_lib_def.define("lsq_backward_per_tensor_wide(Tensor grad, Tensor x, Tensor scale, Tensor shift, " + _TAIL +
                ", int numel_for_scaler) -> (Tensor, Tensor)")
_lib_def.define("lsq_backward_per_channel_wide(Tensor grad, Tensor x, Tensor scale, Tensor shift, int axis, " + _TAIL +
                ", int numel_for_scaler) -> (Tensor, Tensor)")
#  * `lsq_minmax*`: one-pass running min/max (torch.aminmax semantics) for the observer init phase.
#  * `lsq_backward_from_mask`: eval-mode backward from the forward's one-byte inside mask (dx = grad * mask).
_lib_def.define("lsq_backward_from_mask(Tensor grad, Tensor mask) -> Tensor")
_lib_def.define("lsq_minmax_per_tensor(Tensor x) -> (Tensor, Tensor)")
_lib_def.define("lsq_minmax_per_channel(Tensor x, int axis) -> (Tensor, Tensor)")
#  * `lsq_meanstd*`: one-pass mean and unbiased standard deviation (torch.mean / torch.std semantics) for the
#    3-sigma initialisation of weight quantizers.
_lib_def.define("lsq_meanstd_per_tensor(Tensor x) -> (Tensor, Tensor)")
_lib_def.define("lsq_meanstd_per_channel(Tensor x, int axis) -> (Tensor, Tensor)")
_lib_def.define("lsq_quantize_per_tensor(Tensor x, Tensor scale, Tensor shift, int quant_min, int quant_max, "
                "int type_min, int type_max, int level_bias) -> (Tensor, Tensor)")
_lib_def.define("lsq_quantize_per_channel(Tensor x, Tensor scale, Tensor shift, int axis, int quant_min, "
                "int quant_max, int type_min, int type_max, int level_bias) -> (Tensor, Tensor)")
#  * `lsq_levels_*`: ONLY the int8 levels (y is not written: 5 instead of 9 bytes of traffic per fp32 element) -- the
#    conversion-time pass behind `torchlsq.functional.lsq_quantize` / `LSQFakeQuantizer.quantize`, which wraps them into real
#    torch.quint8 / torch.qint8 tensors.  The byte is (q - level_bias) mod 256 (include/lsq_hip.h, lsq_fwd_extras).
_lib_def.define("lsq_levels_per_tensor(Tensor x, Tensor scale, Tensor shift, int quant_min, int quant_max, "
                "int type_min, int type_max, int level_bias) -> Tensor")
_lib_def.define("lsq_levels_per_channel(Tensor x, Tensor scale, Tensor shift, int axis, int quant_min, "
                "int quant_max, int type_min, int type_max, int level_bias) -> Tensor")



# -------------------------------------------------------------------------------------------------
# the HIP backend ("CUDA" dispatch key on ROCm): _hip_host.py
# -------------------------------------------------------------------------------------------------
def _impl_minmax_pt(x):
    return hip_minmax(x, None)


def _impl_minmax_pc(x, axis):
    return hip_minmax(x, axis)


def _impl_meanstd_pt(x):
    return hip_meanstd(x, None)


def _impl_meanstd_pc(x, axis):
    return hip_meanstd(x, axis)


def _impl_fwd_pt(x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode):
    return hip_forward_per_tensor(x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode)


def _impl_bwd_pt(grad, x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode):
    return hip_backward_per_tensor(grad, x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode,
                                   init_mode)


def _impl_fwd_pc(x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode):
    return hip_forward_per_channel(x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode,
                                   init_mode)


def _impl_bwd_pc(grad, x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode):
    return hip_backward_per_channel(grad, x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs, sym,
                                    eval_mode, init_mode)


def _impl_bwd_pt_wide(grad, x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode, n4s):
    return hip_backward_per_tensor(grad, x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode,
                                   init_mode, numel_for_scaler=n4s, want_wide=True)


def _impl_bwd_pc_wide(grad, x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode,
                      n4s):
    return hip_backward_per_channel(grad, x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs, sym,
                                    eval_mode, init_mode, numel_for_scaler=n4s, want_wide=True)


def _impl_quantize_pt(x, scale, shift, qmin, qmax, tmin, tmax, level_bias):
    return hip_forward_per_tensor(x, scale, shift, qmin, qmax, tmin, tmax, True, 1.0, False, False, False,
                                  levels_bias=level_bias)


def _impl_quantize_pc(x, scale, shift, axis, qmin, qmax, tmin, tmax, level_bias):
    return hip_forward_per_channel(x, scale, shift, axis, qmin, qmax, tmin, tmax, True, 1.0, False, False, False,
                                   levels_bias=level_bias)


def _impl_levels_pt(x, scale, shift, qmin, qmax, tmin, tmax, level_bias):
    return hip_forward_per_tensor(x, scale, shift, qmin, qmax, tmin, tmax, True, 1.0, False, False, False,
                                  levels_bias=level_bias, levels_only=True)


def _impl_levels_pc(x, scale, shift, axis, qmin, qmax, tmin, tmax, level_bias):
    return hip_forward_per_channel(x, scale, shift, axis, qmin, qmax, tmin, tmax, True, 1.0, False, False, False,
                                   levels_bias=level_bias, levels_only=True)


_lib_hip = torch.library.Library("torchlsq", "IMPL", "CUDA")
_lib_hip.impl("lsq_levels_per_tensor", _impl_levels_pt)
_lib_hip.impl("lsq_levels_per_channel", _impl_levels_pc)
_lib_hip.impl("lsq_forward_per_tensor", _impl_fwd_pt)
_lib_hip.impl("lsq_backward_per_tensor", _impl_bwd_pt)
_lib_hip.impl("lsq_forward_per_channel", _impl_fwd_pc)
_lib_hip.impl("lsq_backward_per_channel", _impl_bwd_pc)
_lib_hip.impl("lsq_backward_per_tensor_wide", _impl_bwd_pt_wide)
_lib_hip.impl("lsq_backward_per_channel_wide", _impl_bwd_pc_wide)
_lib_hip.impl("lsq_quantize_per_tensor", _impl_quantize_pt)
_lib_hip.impl("lsq_quantize_per_channel", _impl_quantize_pc)
_lib_hip.impl("lsq_backward_from_mask", hip_backward_from_mask)
_lib_hip.impl("lsq_minmax_per_tensor", _impl_minmax_pt)
_lib_hip.impl("lsq_minmax_per_channel", _impl_minmax_pc)
_lib_hip.impl("lsq_meanstd_per_tensor", _impl_meanstd_pt)
_lib_hip.impl("lsq_meanstd_per_channel", _impl_meanstd_pc)



# the CPU backend ("CPU" dispatch key): _cpu_host.py
_lib_cpu = torch.library.Library("torchlsq", "IMPL", "CPU")
_lib_cpu.impl("lsq_forward_per_tensor", lambda x, s, b, *a: cpu_forward(x, s, b, 0, False, *a))
_lib_cpu.impl("lsq_backward_per_tensor", lambda g, x, s, b, *a: cpu_backward(g, x, s, b, 0, False, *a))
_lib_cpu.impl("lsq_forward_per_channel", lambda x, s, b, axis, *a: cpu_forward(x, s, b, axis, True, *a))
_lib_cpu.impl("lsq_backward_per_channel", lambda g, x, s, b, axis, *a: cpu_backward(g, x, s, b, axis, True, *a))
_lib_cpu.impl("lsq_backward_per_tensor_wide",
              lambda g, x, s, b, *a: cpu_backward(g, x, s, b, 0, False, *a[:-1], numel_for_scaler=a[-1], want_wide=True))
_lib_cpu.impl("lsq_backward_per_channel_wide",
              lambda g, x, s, b, axis, *a: cpu_backward(g, x, s, b, axis, True, *a[:-1], numel_for_scaler=a[-1], want_wide=True))
_lib_cpu.impl("lsq_levels_per_tensor", lambda x, s, b, *a: cpu_levels(x, s, b, 0, False, *a))
_lib_cpu.impl("lsq_levels_per_channel", lambda x, s, b, axis, *a: cpu_levels(x, s, b, axis, True, *a))
_lib_cpu.impl("lsq_minmax_per_tensor", lambda x: _cpu_minmax(x))
_lib_cpu.impl("lsq_minmax_per_channel", lambda x, axis: _cpu_minmax(x, axis))
_lib_cpu.impl("lsq_meanstd_per_tensor", lambda x: _cpu_meanstd(x))
_lib_cpu.impl("lsq_meanstd_per_channel", lambda x, axis: _cpu_meanstd(x, axis))


# -------------------------------------------------------------------------------------------------
# shape-only ("meta") kernels so the ops trace under torch.compile / FakeTensor
# -------------------------------------------------------------------------------------------------
def _meta_like(x):
    return torch.empty_like(x)


@torch.library.register_fake("torchlsq::lsq_forward_per_tensor", lib=_lib_def)
def _fake_fwd_pt(x, scale, shift, *a):
    return _meta_like(x)


@torch.library.register_fake("torchlsq::lsq_forward_per_channel", lib=_lib_def)
def _fake_fwd_pc(x, scale, shift, axis, *a):
    return _meta_like(x)


@torch.library.register_fake("torchlsq::lsq_backward_per_tensor", lib=_lib_def)
def _fake_bwd_pt(grad, x, scale, shift, *a):
    return _meta_like(x), scale.new_empty((1,)), shift.new_empty((1,))


@torch.library.register_fake("torchlsq::lsq_backward_per_channel", lib=_lib_def)
def _fake_bwd_pc(grad, x, scale, shift, axis, *a):
    return _meta_like(x), torch.empty_like(scale), torch.empty_like(shift)


@torch.library.register_fake("torchlsq::lsq_backward_per_tensor_wide", lib=_lib_def)
def _fake_bwd_pt_wide(grad, x, scale, shift, *a):
    return _meta_like(x), x.new_empty((2,), dtype=torch.float64)


@torch.library.register_fake("torchlsq::lsq_backward_per_channel_wide", lib=_lib_def)
def _fake_bwd_pc_wide(grad, x, scale, shift, axis, *a):
    return _meta_like(x), x.new_empty((2, scale.numel()), dtype=torch.float64)


@torch.library.register_fake("torchlsq::lsq_quantize_per_tensor", lib=_lib_def)
def _fake_quantize_pt(x, scale, shift, *a):
    return _meta_like(x), torch.empty_like(x, dtype=torch.int8)


@torch.library.register_fake("torchlsq::lsq_quantize_per_channel", lib=_lib_def)
def _fake_quantize_pc(x, scale, shift, axis, *a):
    return _meta_like(x), torch.empty_like(x, dtype=torch.int8)


@torch.library.register_fake("torchlsq::lsq_levels_per_tensor", lib=_lib_def)
def _fake_levels_pt(x, scale, shift, *a):
    return torch.empty_like(x, dtype=torch.int8)


@torch.library.register_fake("torchlsq::lsq_levels_per_channel", lib=_lib_def)
def _fake_levels_pc(x, scale, shift, axis, *a):
    return torch.empty_like(x, dtype=torch.int8)


@torch.library.register_fake("torchlsq::lsq_backward_from_mask", lib=_lib_def)
def _fake_bwd_mask(grad, mask):
    return torch.empty_strided(mask.shape, mask.stride(), dtype=grad.dtype, device=grad.device)


@torch.library.register_fake("torchlsq::lsq_minmax_per_tensor", lib=_lib_def)
def _fake_minmax_pt(x):
    pd = _param_dtype(x)
    return x.new_empty((), dtype=pd), x.new_empty((), dtype=pd)


@torch.library.register_fake("torchlsq::lsq_minmax_per_channel", lib=_lib_def)
def _fake_minmax_pc(x, axis):
    pd = _param_dtype(x)
    return x.new_empty((x.size(axis),), dtype=pd), x.new_empty((x.size(axis),), dtype=pd)


torch.library.register_fake("torchlsq::lsq_meanstd_per_tensor", _fake_minmax_pt, lib=_lib_def)
torch.library.register_fake("torchlsq::lsq_meanstd_per_channel", _fake_minmax_pc, lib=_lib_def)


# -------------------------------------------------------------------------------------------------
# autograd (restates lsq_autograd.cpp: forward saves {input, scale, shift} + the scalars, backward
# calls the backward op through the dispatcher and returns grads for the three tensors only)
# -------------------------------------------------------------------------------------------------
def _setup_pt(ctx, inputs, output):
    x, scale, shift = inputs[:3]
    ctx.save_for_backward(x, scale, shift)
    ctx.lsq_scalars = tuple(inputs[3:])


def _backward_pt(ctx, grad_out):
    x, scale, shift = ctx.saved_tensors
    dx, ds, db = torch.ops.torchlsq.lsq_backward_per_tensor(grad_out, x, scale, shift, *ctx.lsq_scalars)
    return (dx, ds, db) + (None,) * 9      # lsq_autograd.cpp:69-71


def _setup_pc(ctx, inputs, output):
    x, scale, shift = inputs[:3]
    ctx.save_for_backward(x, scale, shift)
    ctx.lsq_scalars = tuple(inputs[3:])  # axis first


def _backward_pc(ctx, grad_out):
    x, scale, shift = ctx.saved_tensors
    dx, ds, db = torch.ops.torchlsq.lsq_backward_per_channel(grad_out, x, scale, shift, *ctx.lsq_scalars)
    return (dx, ds, db) + (None,) * 10     # lsq_autograd.cpp:169-170


torch.library.register_autograd("torchlsq::lsq_forward_per_tensor", _backward_pt, setup_context=_setup_pt,
                                lib=_lib_def)
torch.library.register_autograd("torchlsq::lsq_forward_per_channel", _backward_pc, setup_context=_setup_pc,
                                lib=_lib_def)


def _no_double_backward(name):
    def bw(ctx, *grads):
        raise RuntimeError("double backwards on %s not supported" % name)  # lsq_autograd.cpp:106,208
    return bw


for _op, _nm in (("lsq_backward_per_tensor", "lsq_per_tensor"), ("lsq_backward_per_channel", "lsq_per_channel"),
                 ("lsq_backward_per_tensor_wide", "lsq_per_tensor"),
                 ("lsq_backward_per_channel_wide", "lsq_per_channel")):
    torch.library.register_autograd("torchlsq::" + _op, _no_double_backward(_nm), lib=_lib_def)


# -------------------------------------------------------------------------------------------------
# composite front op (restates quantops::ops::lsq, lsq.cpp:104-134) and _cuda_version
# -------------------------------------------------------------------------------------------------
def _lsq_front(x, scale, shift, quant_min, quant_max, type_min, type_max, axis, use_grad_scaling, grad_scale,
               is_affine, is_perchannel, eval_mode, init_mode):
    _check(scale.dim() == 1,
           "scale should be a 1-D tensor, even in per tensor case(please, avoid torch.Scalar too)")
    _check(shift.dim() == 1,
           "shift should be a 1-D tensor, even in per tensor case(please, avoid torch.Scalar too)")
    sym = not is_affine
    if is_perchannel:
        # a size-1 parameter is repeated up to the larger size; `repeat` is differentiable, so the
        # size-1 leaf still receives the summed gradient (lsq.cpp:124-126)
        size = max(scale.size(0), shift.size(0))
        _scale = scale if scale.size(0) == size else scale.repeat(size)
        _shift = shift if shift.size(0) == size else shift.repeat(size)
        return torch.ops.torchlsq.lsq_forward_per_channel(x, _scale, _shift, axis, quant_min, quant_max, type_min,
                                                          type_max, use_grad_scaling, grad_scale, sym, eval_mode,
                                                          init_mode)
    return torch.ops.torchlsq.lsq_forward_per_tensor(x, scale, shift, quant_min, quant_max, type_min, type_max,
                                                     use_grad_scaling, grad_scale, sym, eval_mode, init_mode)


_lib_def.impl("lsq", _lsq_front, "CompositeImplicitAutograd")


def _runtime_version():
    """torchlsq::_cuda_version of this build: the HIP_VERSION the kernels were compiled with, or -1
    when the native library is missing (reference torchlsq.cpp:25-31 returns CUDA_VERSION / -1)."""
    return int(_abi._LIB.lsq_hip_runtime_version()) if _abi._HAS_OPS else -1


_lib_def.impl("_cuda_version", _runtime_version, "CompositeExplicitAutograd")

_check_hip_version()
