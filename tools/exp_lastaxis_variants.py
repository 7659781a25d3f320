#!/usr/bin/env python3
"""Sweep unroll x workgroups/CU x pipelined of the WINDOW-mode kernels on last-axis (inner = 1) per-channel shapes (tuning build)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from torchlsq import synth
from torchlsq.extension import C_ABI, LsqParams
import lsq_tools
C_ABI_INTERNAL = lsq_tools.internal_abi()
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_tune", "liblsq_hip_tune.so"))
for tbl in (C_ABI, C_ABI_INTERNAL):
    for name, (res, args) in tbl.items():
        getattr(lib, name).restype = res; getattr(lib, name).argtypes = args
dev = torch.device("cuda:0")


def timeit(fn, reps=10):
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
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


for dt, code in ((torch.bfloat16, 2), (torch.float32, 0)):
    for outer, C in ((8192, 4096), (65536, 1024), (12608, 768), (200704, 256)):
        n = outer * C
        x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt)
        g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt)
        scale = synth.uniform_like(C, 3, 0.02, 0.05, device=dev); shift = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        y = torch.empty_like(x); dx = torch.empty_like(x)
        ds = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
        ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
        p = LsqParams(0, 127, 0, 255, 1, 0, 0, 0, 1.0, 0)
        rows = []
        for unroll in (1, 2, 4, 8):
            for bpc in (2, 4, 8, 16):
                v = unroll | (1 << 8) | (1 << 9) | (bpc << 16)
                def fwd(s):
                    assert lib.lsq_hip_forward_per_channel_ex(code, x.data_ptr(), y.data_ptr(), outer, C, 1, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None, s, v) == 0
                def bwd(s):
                    assert lib.lsq_hip_backward_per_channel_ex(code, g.data_ptr(), x.data_ptr(), dx.data_ptr(), ds.data_ptr(), db.data_ptr(), None, outer, C, 1, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None, ws.data_ptr(), ws.numel(), s, v) == 0
                tf, tb = timeit(fwd), timeit(bwd)
                v |= 1 << 10
                tbp = timeit(bwd)
                rows.append((unroll, bpc, round(tf, 1), round(tb, 1), round(tbp, 1)))
        print(str(dt).replace("torch.", ""), (outer, C), "fwd best:", sorted(rows, key=lambda r: r[2])[:3], "| bwd plain best:",
              sorted(rows, key=lambda r: r[3])[:3], "| bwd pipelined best:", sorted(rows, key=lambda r: r[4])[:3])
        print("    all (unroll, wg/CU, fwd, bwd plain, bwd pipelined):", rows)
