#!/usr/bin/env python3
"""Sweep unroll x workgroups-per-CU of the SEGMENT-mode per-channel kernels on weight shapes (tuning build, diagnostic)."""
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
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


for dt, code in ((torch.bfloat16, 2), (torch.float32, 0)):
    for shape in ((512, 512, 3, 3), (4096, 4096), (4096, 11008), (32000, 4096), (1024, 1024, 3, 3)):
        n = 1
        for d in shape: n *= d
        C, inner = shape[0], n // shape[0]
        x = synth.normal_like(n, 1, 0.0, 0.05, device=dev, dtype=dt)
        g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt)
        scale = synth.uniform_like(C, 3, 5e-4, 2.5e-3, device=dev); shift = torch.zeros(C, device=dev)
        y = torch.empty_like(x); dx = torch.empty_like(x)
        ds = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
        ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        p = LsqParams(-128, 127, -128, 127, 1, 1, 0, 0, 1.0, 0)
        esz = x.element_size()
        rows = []
        for unroll in (1, 2, 4, 8):
            for bpc in (4, 8, 16):
                v = unroll | (1 << 8) | (1 << 9) | (bpc << 16)
                def fwd(s):
                    assert lib.lsq_hip_forward_per_channel_ex(code, x.data_ptr(), y.data_ptr(), 1, C, inner, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None, s, v) == 0
                def bwd(s):
                    assert lib.lsq_hip_backward_per_channel_ex(code, g.data_ptr(), x.data_ptr(), dx.data_ptr(), ds.data_ptr(), db.data_ptr(), None, 1, C, inner, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None, ws.data_ptr(), ws.numel(), s, v) == 0
                rows.append((unroll, bpc, round(timeit(fwd), 2), round(timeit(bwd), 2)))
        print(str(dt).replace("torch.", ""), shape, "fwd best:", sorted(rows, key=lambda r: r[2])[:3], "| bwd best:", sorted(rows, key=lambda r: r[3])[:3])
        print("    all (unroll, wg/CU, fwd_us, bwd_us):", rows)
