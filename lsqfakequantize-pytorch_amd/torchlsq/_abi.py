"""The native libraries of the package and how they are opened: liblsq_hip.so (include/lsq_hip.h, ctypes), liblsq_cpu.so
(include/lsq_cpu.h, ctypes) and _lsq_torch.so (the C++ torch binding of the same C ABI, torch.ops.load_library).

This is the replacement of reference torchlsq/extension.py:12-56, which located `_C.so` and `torch.ops.load_library`-ed it.
The module holds the loader STATE (`_LIB`, `_HAS_OPS`, `_CPU_LIB`, `_NATIVE_LSQ`); the host layers (_hip_host.py,
_cpu_host.py) read it through the module at call time, so tools/lsq_tools.py can swap the tools build of the library in with
`set_library`.
"""
import ctypes
import os
import threading

import torch

_HAS_OPS = False
error_str = ""
_LIB = None
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblsq_hip.so")

# dtype codes of include/lsq_hip.h
LSQ_F32, LSQ_F64, LSQ_BF16, LSQ_F16 = 0, 1, 2, 3
_DTYPE_CODE = {torch.float32: LSQ_F32, torch.float64: LSQ_F64, torch.bfloat16: LSQ_BF16, torch.float16: LSQ_F16}


class LsqParams(ctypes.Structure):
    """struct lsq_params (include/lsq_hip.h)."""
    _fields_ = [("quant_min", ctypes.c_int32), ("quant_max", ctypes.c_int32),
                ("type_min", ctypes.c_int32), ("type_max", ctypes.c_int32),
                ("use_grad_scaling", ctypes.c_int32), ("sym", ctypes.c_int32),
                ("eval_mode", ctypes.c_int32), ("init_mode", ctypes.c_int32),
                ("grad_scaler", ctypes.c_double), ("numel_for_scaler", ctypes.c_int64)]


class LsqFwdExtras(ctypes.Structure):
    """struct lsq_fwd_extras (include/lsq_hip.h)."""
    _fields_ = [("levels", ctypes.c_void_p), ("level_bias", ctypes.c_int32), ("aux_kind", ctypes.c_int32)]


class LsqBwdExtras(ctypes.Structure):
    """struct lsq_bwd_extras (include/lsq_hip.h)."""
    _fields_ = [("ticket", ctypes.c_void_p)]


class LsqObserverUpdate(ctypes.Structure):
    """struct lsq_observer_update (include/lsq_hip.h)."""
    _fields_ = [("mode", ctypes.c_int32), ("first", ctypes.c_int32), ("averaging_constant", ctypes.c_float),
                ("quant_min", ctypes.c_int32), ("quant_max", ctypes.c_int32), ("symmetric", ctypes.c_int32),
                ("zero_point_symmetric", ctypes.c_int32), ("eps", ctypes.c_float)]


class LsqPcItem(ctypes.Structure):
    """struct lsq_pc_item (include/lsq_hip.h): one tensor of a multi-tensor launch."""
    _fields_ = [("x", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("y", ctypes.c_void_p), ("dx", ctypes.c_void_p),
                ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("ds", ctypes.c_void_p), ("db", ctypes.c_void_p),
                ("outer", ctypes.c_int64), ("channels", ctypes.c_int64), ("inner", ctypes.c_int64)]


class LsqCommOptions(ctypes.Structure):
    """struct lsq_comm_options (include/lsq_hip.h)."""
    _fields_ = [("size", ctypes.c_int32), ("event_system_fence", ctypes.c_int32), ("reserved", ctypes.c_int32 * 2)]


LSQ_TICKET_BYTES = 4096
ABI_VERSION = 6

_vp, _i64, _int, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
_PP = ctypes.POINTER(LsqParams)
_EP = ctypes.POINTER(LsqFwdExtras)
_BP = ctypes.POINTER(LsqBwdExtras)

# every symbol include/lsq_hip.h declares: (restype, argtypes)
C_ABI = {
    "lsq_hip_abi_version": (_int, []),
    "lsq_hip_runtime_version": (_i64, []),
    "lsq_hip_last_error": (ctypes.c_char_p, []),
    "lsq_hip_grad_scaler": (ctypes.c_double, [_int, _int, _i64, ctypes.c_int32, _i64, ctypes.c_int32, ctypes.c_double]),
    "lsq_hip_policy_ticket": (_int, [ctypes.c_int32, ctypes.c_int32, _i64]),
    "lsq_hip_policy_saves_mask": (_int, [ctypes.c_int32] * 4),
    "lsq_hip_backward_per_tensor_workspace": (_sz, [_int, _i64]),
    "lsq_hip_forward_per_tensor": (_int, [_int, _vp, _vp, _i64, _vp, _vp, _PP, _EP, _vp]),
    "lsq_hip_backward_per_tensor": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _PP, _BP, _vp, _sz, _vp]),
    "lsq_hip_backward_per_channel_workspace": (_sz, [_int, _i64, _i64, _i64]),
    "lsq_hip_forward_per_channel": (_int, [_int, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _PP, _EP, _vp]),
    "lsq_hip_backward_per_channel": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _PP, _BP,
                                            _vp, _sz, _vp]),
    "lsq_hip_sharded_finish": (_int, [_int, _vp, _i64, ctypes.c_int32, _PP, _vp, _vp, _vp]),
    "lsq_hip_per_channel_multi_ok": (_int, [_int, _i64, _i64, _i64, _int]),
    "lsq_hip_plan_backward_per_channel": (_int, [_int, _i64, _i64, _i64, _int, _PP, ctypes.POINTER(ctypes.c_int32 * 8)]),
    "lsq_hip_comm_unique_id": (_int, [_vp]),
    "lsq_hip_comm_create": (_int, [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(LsqCommOptions), ctypes.POINTER(_vp)]),
    "lsq_hip_comm_configure": (_int, [_vp, ctypes.POINTER(LsqCommOptions)]),
    "lsq_hip_comm_tune": (_int, [_vp, _vp]),
    "lsq_hip_comm_destroy": (_int, [_vp]),
    "lsq_hip_comm_info": (_int, [_vp, ctypes.POINTER(ctypes.c_int32 * 8)]),
    "lsq_hip_comm_side_stream": (_vp, [_vp]),
    "lsq_hip_comm_join": (_int, [_vp, _vp]),
    "lsq_hip_comm_all_reduce": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp]),
    "lsq_hip_comm_all_reduce_begin": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, ctypes.POINTER(ctypes.c_int32)]),
    "lsq_hip_comm_all_reduce_end": (_int, [_vp, ctypes.c_int32, _vp]),
    "lsq_hip_forward_per_channel_multi": (_int, [_int, ctypes.POINTER(LsqPcItem), ctypes.c_int32, _PP, _vp]),
    "lsq_hip_backward_per_channel_multi": (_int, [_int, ctypes.POINTER(LsqPcItem), ctypes.c_int32, _PP, _vp]),
    "lsq_hip_relayout": (_int, [_int, _vp, _vp, _i64, _i64, _i64, _vp]),
    "lsq_hip_backward_from_mask": (_int, [_int, _vp, _vp, _vp, _i64, _vp]),
    "lsq_hip_minmax_workspace": (_sz, [_int, _i64, _i64, _i64]),
    "lsq_hip_minmax_per_tensor": (_int, [_int, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "lsq_hip_minmax_per_channel": (_int, [_int, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "lsq_hip_meanstd_workspace": (_sz, [_int, _i64, _i64, _i64]),
    "lsq_hip_meanstd_per_tensor": (_int, [_int, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "lsq_hip_meanstd_per_channel": (_int, [_int, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "lsq_hip_observer_update": (_int, [_i64, _vp, _vp, _vp, _vp, ctypes.POINTER(LsqObserverUpdate), _vp, _vp, _vp]),
}
# NOT bound here: the `_ex` twins (a trailing launch-variant code) and the lsq_hip_debug_* knobs of csrc/lsq_internal.h.
# They exist only in the tools build of the library (tools/_tune/liblsq_hip_tools.so), which tools/lsq_tools.py loads and
# swaps in for this module's handle; the `variant` arguments below are for that build and raise on the production library.


def _load_library():
    """dlopen liblsq_hip.so and type its entry points (the replacement of reference extension.py:39-45)."""
    global _LIB
    if not os.path.isfile(_LIB_PATH):
        raise ImportError("%s not found -- build it with `python __graft_entry__.py` or "
                          "`make -C lsqfakequantize-pytorch_amd/csrc`" % _LIB_PATH)
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in C_ABI.items():
        fn = getattr(lib, name)  # AttributeError -> OSError-like failure below
        fn.restype = res
        fn.argtypes = args
    abi = lib.lsq_hip_abi_version()
    if abi != ABI_VERSION:
        raise ImportError("liblsq_hip.so has ABI version %d, this package needs %d" % (abi, ABI_VERSION))
    _LIB = lib


try:
    _load_library()
    _HAS_OPS = True
except (ImportError, OSError, AttributeError) as e:  # surfaced by _assert_has_ops(), like the reference
    error_str = str(e)


# The kernels for tensors in host memory (include/lsq_cpu.h): the counterpart of the reference's CPU dispatch.
_CPU_LIB = None
_CPU_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblsq_cpu.so")
cpu_error_str = ""
C_ABI_CPU = {
    "lsq_cpu_abi_version": (_int, []),
    "lsq_cpu_last_error": (ctypes.c_char_p, []),
    "lsq_cpu_set_num_threads": (None, [_int]),
    "lsq_cpu_forward_per_tensor": (_int, [_int, _vp, _vp, _i64, _vp, _vp, _PP]),
    "lsq_cpu_backward_per_tensor": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _PP]),
    "lsq_cpu_forward_per_channel": (_int, [_int, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _PP]),
    "lsq_cpu_backward_per_channel": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _PP]),
    "lsq_cpu_sharded_finish": (_int, [_int, _vp, _i64, ctypes.c_int32, _PP, _vp, _vp]),
}


def _load_cpu_library():
    global _CPU_LIB, cpu_error_str
    try:
        lib = ctypes.CDLL(_CPU_LIB_PATH)
        for name, (res, args) in C_ABI_CPU.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.lsq_cpu_abi_version() != ABI_VERSION:
            raise OSError("liblsq_cpu.so was built for another ABI version")
        _CPU_LIB = lib
    except (OSError, AttributeError) as e:
        cpu_error_str = str(e)


_load_cpu_library()


# The optional second host layer: torchlsq/_lsq_torch.so, the C++ torch binding of the same C ABI
# (csrc/torch_binding/lsq_torch_binding.cpp, namespace `torchlsq_native`).  It adds no device code; it only
# moves the per-call tensor bookkeeping and the autograd node from Python to C++.  `functional.lsq` prefers it
# for GPU tensors; everything in this module keeps working without it.  TORCHLSQ_HOST_BINDING=ctypes skips it.
_NATIVE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lsq_torch.so")
_NATIVE_LSQ = None
native_error_str = ""


def _load_native_binding():
    global _NATIVE_LSQ, native_error_str
    if os.environ.get("TORCHLSQ_HOST_BINDING", "").lower() == "ctypes":
        native_error_str = "disabled by TORCHLSQ_HOST_BINDING=ctypes"
        return
    if not os.path.isfile(_NATIVE_PATH):
        native_error_str = "%s not found (make -C lsqfakequantize-pytorch_amd/csrc binding)" % _NATIVE_PATH
        return
    try:
        torch.ops.load_library(_NATIVE_PATH)
        if int(torch.ops.torchlsq_native._abi_version()) != ABI_VERSION:
            raise OSError("_lsq_torch.so was built against another ABI version of liblsq_hip.so")
        _NATIVE_LSQ = torch.ops.torchlsq_native.lsq.default
    except (OSError, RuntimeError, AttributeError) as e:
        native_error_str = str(e)


if _HAS_OPS:
    _load_native_binding()


def native_lsq():
    """`torch.ops.torchlsq_native.lsq` (the C++ front op + autograd node) or None when _lsq_torch.so is absent."""
    return _NATIVE_LSQ


def host_binding():
    """'native' (C++ torch binding loaded) or 'ctypes'."""
    return "native" if _NATIVE_LSQ is not None else "ctypes"


def set_host_binding(kind):
    """Switch `functional.lsq` between the two host layers at run time (tests, A/B measurements)."""
    global _NATIVE_LSQ
    if kind == "ctypes":
        _NATIVE_LSQ = None
    elif kind == "native":
        if not hasattr(torch.ops, "torchlsq_native") or not os.path.isfile(_NATIVE_PATH):
            raise RuntimeError("the C++ torch binding is not available: %s" % native_error_str)
        try:
            _NATIVE_LSQ = torch.ops.torchlsq_native.lsq.default
        except (AttributeError, RuntimeError):
            torch.ops.load_library(_NATIVE_PATH)
            _NATIVE_LSQ = torch.ops.torchlsq_native.lsq.default
    else:
        raise ValueError("host binding must be 'native' or 'ctypes'")


def _has_ops():
    return _HAS_OPS


def _assert_has_ops():
    if not _HAS_OPS:
        raise RuntimeError(
            "torchlsq (MI355X build): the native HIP library could not be loaded, so the LSQ ops are "
            "unavailable.  There is no CPU or eager fallback.  Build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950).\n\nImport error details:\n\t%s" % error_str)


def library():
    """The ctypes handle of liblsq_hip.so (raises if it is missing)."""
    _assert_has_ops()
    return _LIB


def _check_hip_version():
    """Counterpart of the reference's _check_cuda_version (extension.py:71-96): the HIP runtime the
    library was compiled against must have the same major version as the one PyTorch uses."""
    if not _HAS_OPS:
        return -1
    v = int(_LIB.lsq_hip_runtime_version())
    hip = getattr(torch.version, "hip", None)
    if v > 0 and hip is not None:
        lib_major = v // 10000000
        t_major = int(hip.split(".")[0])
        if lib_major != t_major:
            raise RuntimeError("Detected that PyTorch and torchlsq were compiled with different HIP versions. "
                               "PyTorch has HIP Version=%s and torchlsq has HIP_VERSION=%d. "
                               "Please rebuild torchlsq against your PyTorch's ROCm." % (hip, v))
    return v

def set_library(lib):
    """Make `lib` (a typed ctypes handle exporting include/lsq_hip.h) the library the Python host layer calls; returns the
    previous handle.  For tools/lsq_tools.py (the tools build) -- the C++ binding keeps the library it was linked against."""
    global _LIB
    prev, _LIB = _LIB, lib
    return prev
