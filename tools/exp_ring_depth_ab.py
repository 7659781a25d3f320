#!/usr/bin/env python3
"""A/B on one box: depth of the backward's LDS-DMA ring (rows in flight per wave): 4 (shipped) against 6 and 8 (experiment
builds tools/_tune/liblsq_hip_depth{6,8}.so, -DLSQ_BWD_DMA_DEPTH=N); default launch policy; GPU-side us per backward."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch
import torchlsq  # noqa: F401
from torchlsq import extension as E, synth
from torchlsq.extension import C_ABI, LsqParams

dev = torch.device("cuda:0")
libs = {"4": E.library()}
for d in ("6", "8"):
    libs[d] = ctypes.CDLL(os.path.join(ROOT, "tools", "_tune", "liblsq_hip_depth%s.so" % d))
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


for (outer, C, inner) in ((256, 2048, 49), (12608, 768, 1), (8192, 4096, 1), (32, 256, 3136), (200704, 256, 1), (65536, 1024, 1)):
    for dt, code in ((torch.float32, 0), (torch.bfloat16, 2)):
        n = outer * C * inner
        x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt)
        g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt)
        scale = synth.uniform_like(C, 3, 0.02, 0.05, device=dev); shift = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        p = LsqParams(0, 127, 0, 255, 1, 0, 0, 0, 1.0, 0)
        ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
        dx = torch.empty_like(x); ds = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
        res = {}
        for rnd in range(2):
            for name, lib in libs.items():
                def bwd(s, lib=lib):
                    assert lib.lsq_hip_backward_per_channel(code, g.data_ptr(), x.data_ptr(), dx.data_ptr(), ds.data_ptr(), db.data_ptr(), None,
                                                            outer, C, inner, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None,
                                                            ws.data_ptr(), ws.numel(), s) == 0
                res.setdefault(name, []).append(timeit(bwd))
        print("%-8s [%d,%d,%d] bwd us: %s" % (str(dt).replace("torch.", ""), outer, C, inner,
              "  ".join("depth %s %.1f" % (k, min(v)) for k, v in res.items())), flush=True)
