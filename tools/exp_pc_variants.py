#!/usr/bin/env python3
"""Sweep unroll x workgroups-per-CU of the window-mode per-channel kernels with the TUNING library (diagnostic)."""
import ctypes, os, sys, json
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
stream = torch.cuda.current_stream().cuda_stream

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

# usage: exp_pc_variants.py [--eval] [outer C inner]   (default: BASELINE config 5 = 256 2048 49)
EVAL = "--eval" in sys.argv      # the dx-only backward: the streaming rate of the access pattern, almost no arithmetic
_args = [a for a in sys.argv[1:] if a != "--eval"]
GEOM = tuple(int(v) for v in _args[:3]) if len(_args) >= 3 else (256, 2048, 49)
for dt, code in ((torch.float32, 0), (torch.bfloat16, 2)):
    outer, C, inner = GEOM
    n_ = outer * C * inner
    x = synth.normal_like(n_, 1, 0.0, 1.0, device=dev, dtype=dt)
    g = synth.normal_like(n_, 2, 0.0, 1e-3, device=dev, dtype=dt)
    scale = synth.uniform_like(C, 3, 0.05, 0.35, device=dev)
    shift = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    ds = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    p = LsqParams(-8, 7, -128, 127, 1, 0, 1 if EVAL else 0, 0, 1.0, 0)
    n = x.numel(); esz = x.element_size()
    rows = []
    for unroll in (1, 2, 4, 8):
        for bpc in (1, 2, 3, 4, 6, 8, 16):
            v = unroll | (1 << 8) | (1 << 9) | (bpc << 16)
            def fwd(s):
                assert lib.lsq_hip_forward_per_channel_ex(code, x.data_ptr(), y.data_ptr(), outer, C, inner, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None, s, v) == 0
            def bwd(s):
                assert lib.lsq_hip_backward_per_channel_ex(code, g.data_ptr(), x.data_ptr(), dx.data_ptr(), ds.data_ptr(), db.data_ptr(), None, outer, C, inner, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None, ws.data_ptr(), ws.numel(), s, v) == 0
            tf, tb = timeit(fwd), timeit(bwd)
            v |= 1 << 10                      # the software-pipelined backward loop (tuning builds: `chunked` bit)
            tbp = timeit(bwd)
            rows.append((unroll, bpc, round(tf, 2), round(2 * esz * n / tf / 1e3), round(tb, 2), round(3 * esz * n / tb / 1e3),
                         round(tbp, 2)))
    print(str(dt), "best fwd:", sorted(rows, key=lambda r: r[2])[:4])
    print(str(dt), "best bwd:", sorted(rows, key=lambda r: r[4])[:6])
    print(str(dt), "best bwd pipelined:", sorted([(r[0], r[1], r[6]) for r in rows], key=lambda r: r[2])[:6])
    print(str(dt), "bwd (unroll, wg/CU, plain us, pipelined us):", [(r[0], r[1], r[4], r[6]) for r in rows if r[1] in (2, 4, 8, 16)])
