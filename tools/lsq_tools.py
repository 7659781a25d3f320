"""The TOOLS build of the HIP library for the scripts under tools/ and the branch-pinning tests.

    import lsq_tools
    lib = lsq_tools.activate()        # tools/_tune/liblsq_hip_tools.so, typed; also swapped in for torchlsq.extension's handle
    lib.lsq_hip_debug_set_ring_nt(1)
    ...
    lsq_tools.deactivate()            # back to the production library

`liblsq_hip.so` exports exactly include/lsq_hip.h.  What tuning sweeps and A/B measurements need on top -- the `_ex` twins
of the four ops (a trailing launch-variant code), the lsq_hip_debug_* policy knobs and the launch note of
csrc/lsq_internal.h -- lives in a second build of the same sources with -DLSQ_TOOLS (`make -C lsqfakequantize-pytorch_amd/csrc
tools`, part of `python __graft_entry__.py`).  `activate()` loads it, types every entry point and makes it the library
`torchlsq.extension`'s Python host layer calls, so `extension.hip_*(..., variant=V)`, `extension.library()` and everything
above them (functional.lsq through the ctypes binding, the modules) run on it.  The C++ host binding (_lsq_torch.so) is
linked against the production library and cannot see the knobs: activate() switches the host layer to ctypes.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
TOOLS_LIB = os.path.join(ROOT, "tools", "_tune", "liblsq_hip_tools.so")

_state = {"lib": None, "saved": None}


def internal_abi():
    """name -> (restype, argtypes) of csrc/lsq_internal.h"""
    from torchlsq.extension import C_ABI
    _int = ctypes.c_int
    tab = {n + "_ex": (C_ABI[n][0], C_ABI[n][1] + [_int]) for n in
           ("lsq_hip_forward_per_tensor", "lsq_hip_backward_per_tensor", "lsq_hip_forward_per_channel",
            "lsq_hip_backward_per_channel")}
    for n in ("force_ring", "set_ww_min_rows", "set_ww_split64", "set_ww_big", "set_ring_nt", "set_fin_ch",
              "set_observe_wg_per_cu", "set_ww_max_log2", "set_seg_min_div", "set_fwd_direct", "set_seg_no_up_front", "set_own", "set_own_min_run", "set_own_fat"):
        tab["lsq_hip_debug_" + n] = (None, [_int])
    tab["lsq_hip_debug_last_launch"] = (None, [ctypes.POINTER(ctypes.c_int * 8)])
    return tab


def load(path=TOOLS_LIB):
    """dlopen the tools library (no side effect on torchlsq.extension) and type include/lsq_hip.h + csrc/lsq_internal.h."""
    from torchlsq.extension import C_ABI
    if not os.path.isfile(path):
        raise ImportError("%s not found -- build it with `make -C lsqfakequantize-pytorch_amd/csrc tools`" % path)
    lib = ctypes.CDLL(path)
    for table in (C_ABI, internal_abi()):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    return lib


def activate(path=TOOLS_LIB):
    from torchlsq import extension as E
    if _state["lib"] is None:
        _state["lib"] = load(path)
    if _state["saved"] is None:
        _state["saved"] = (E.library(), E.host_binding())
    E.set_library(_state["lib"])
    _drop_memos()                    # the tools build sizes the scratch for every variant: do not reuse production answers
    E.set_host_binding("ctypes")
    return _state["lib"]


def _drop_memos():
    """per-shape answers the Python host layer keeps (workspace sizes, multi-tensor eligibility) depend on the library and on
    its policy knobs (set_seg_min_div, ...): forget them whenever either changes"""
    from torchlsq import _hip_host
    _hip_host._WS_BYTES_PC.clear()
    _hip_host._MULTI_OK.clear()


def set_knob(name, value):
    """lsq_hip_debug_<name>(value) on the active tools library + forget the host layer's memoised policy answers"""
    getattr(_state["lib"], "lsq_hip_debug_" + name)(int(value))
    _drop_memos()


def deactivate():
    from torchlsq import extension as E
    if _state["saved"] is not None:
        reset_knobs()
        E.set_library(_state["saved"][0])
        _drop_memos()
        if _state["saved"][1] == "native":
            E.set_host_binding("native")
        _state["saved"] = None


def reset_knobs():
    lib = _state["lib"]
    if lib is not None:
        for n in ("force_ring", "set_ww_min_rows", "set_ww_split64", "set_ww_big", "set_ring_nt", "set_fin_ch",
                  "set_observe_wg_per_cu", "set_ww_max_log2", "set_seg_min_div", "set_fwd_direct", "set_seg_no_up_front", "set_own", "set_own_min_run", "set_own_fat"):
            getattr(lib, "lsq_hip_debug_" + n)(0)
        _drop_memos()


KINDS = {0: "none", 1: "windows", 2: "row-groups", 3: "segment", 4: "owners"}


def last_launch():
    """The calling thread's last per-channel backward launch on the tools library, as a dict."""
    out = (ctypes.c_int * 8)()
    _state["lib"].lsq_hip_debug_last_launch(ctypes.byref(out))
    return dict(grid_x=out[0], grid_y=out[1], resident_per_cu=out[2], vgprs=out[3], kind=KINDS.get(out[4], "?"),
                ring_depth=out[5], block=out[6], ring_nt=out[7])
