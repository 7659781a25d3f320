#!/usr/bin/env python3
"""A/B on one box: 16-bit window-mode backward with fp32 pre-sums of 4 rows before the fp64 accumulation (experiment
build tools/_tune/liblsq_hip_pre32.so, -DLSQ_PRE32) against the shipped library (a convert + fp64 add per term)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch
import torchlsq  # noqa: F401
from torchlsq import extension as E, synth
from torchlsq.extension import C_ABI, LsqParams

dev = torch.device("cuda:0")
libs = {"shipped": E.library(), "pre32": ctypes.CDLL(os.path.join(ROOT, "tools", "_tune", "liblsq_hip_pre32.so"))}
for lib in libs.values():
    for name, (res, args) in C_ABI.items():
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args


def timeit(fn, reps=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        s = st.cuda_stream
        fn(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps):
                fn(s)
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


for (outer, C, inner) in ((256, 2048, 49), (32, 256, 3136), (8192, 4096, 1), (64, 64, 12544), (200704, 256, 1)):
    n = outer * C * inner
    x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=torch.bfloat16)
    g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=torch.bfloat16)
    scale = synth.uniform_like(C, 3, 0.02, 0.05, device=dev); shift = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
    p = LsqParams(0, 127, 0, 255, 1, 0, 0, 0, 1.0, 0)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    out = {}
    res = {}
    for name, lib in libs.items():
        dx = torch.empty_like(x); ds = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
        def bwd(s, lib=lib, dx=dx, ds=ds, db=db):
            assert lib.lsq_hip_backward_per_channel(2, g.data_ptr(), x.data_ptr(), dx.data_ptr(), ds.data_ptr(), db.data_ptr(), None,
                                                    outer, C, inner, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None,
                                                    ws.data_ptr(), ws.numel(), s) == 0
        res[name] = [timeit(bwd) for _ in range(3)]
        out[name] = (dx.clone(), ds.clone(), db.clone())
    same = torch.equal(out["shipped"][0], out["pre32"][0])
    rel = max(float(((out["shipped"][k] - out["pre32"][k]).abs() / out["shipped"][k].abs().clamp_min(1e-30)).max()) for k in (1, 2))
    a, b = min(res["shipped"]), min(res["pre32"])
    print("bf16 [%d,%d,%d] bwd: shipped %.1f us | pre32 %.1f us (%+.1f%%) | dx %s, ds/db max rel diff %.1e" %
          (outer, C, inner, a, b, (b / a - 1) * 100, "same" if same else "DIFFERENT", rel), flush=True)
