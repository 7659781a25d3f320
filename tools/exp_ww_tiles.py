#!/usr/bin/env python3
"""Row-group-window backward, 16-bit and 4-byte storage, quantized axis last: time against rows walked per workgroup
(the rows-per-workgroup floor is the knob), usual and 768/1024-lane workgroups -- separates the per-workgroup cost from the
per-row cost.  GPU-side us per backward incl. finalize (HIP graph of 20), (windows x splits) in brackets."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torchlsq  # noqa: F401
from torchlsq import extension as E, synth
import lsq_tools  # noqa: E402  (tools build of the library: `_ex` entry points, lsq_hip_debug_* knobs)
lsq_tools.activate()
lib = E.library()
lib.lsq_hip_debug_set_ww_big.argtypes = [ctypes.c_int]
lib.lsq_hip_debug_set_ww_min_rows.argtypes = [ctypes.c_int]
lib.lsq_hip_debug_last_launch.argtypes = [ctypes.POINTER(ctypes.c_int * 8)]
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps):
                fn()
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


for shape in ((3152, 768), (12608, 768), (4096, 1024)):
    for dt in (torch.bfloat16, torch.float32):
        n = shape[0] * shape[1]
        x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt).view(shape)
        g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt).view(shape)
        s = synth.uniform_like(shape[1], 3, 0.02, 0.05, device=dev); b = synth.normal_like(shape[1], 4, 0.0, 0.1, device=dev)
        q = (0, 127, 0, 255, True, 1.0, False, False, False)
        for big in (2, 1):
            res = []
            for mr in (2, 4, 8, 16, 32, 64, 128, 256):
                lib.lsq_hip_debug_set_ww_big(big); lib.lsq_hip_debug_set_ww_min_rows(mr)
                E._WS_BYTES_PC.clear()
                t = timeit(lambda: E.hip_backward_per_channel(g, x, s, b, 1, *q))
                o = (ctypes.c_int * 8)(); lib.lsq_hip_debug_last_launch(ctypes.byref(o))
                res.append("%d: %.1f (%dx%d)" % (mr, t, o[0], o[1]))
            print("%-9s %-14s %-5s %s" % (str(dt).replace("torch.", ""), shape, "big" if big == 1 else "usual", "  ".join(res)), flush=True)
lib.lsq_hip_debug_set_ww_big(0); lib.lsq_hip_debug_set_ww_min_rows(0)
