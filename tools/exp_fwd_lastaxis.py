#!/usr/bin/env python3
"""Forward on last-axis (inner = 1) per-channel shapes: the default launch against forced register loops / ring at several
workgroups-per-CU (variant bits 16-23, bits 12-13: 1 = registers, 2 = ring).  GPU-side us per forward (HIP graph of 20)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torchlsq  # noqa: F401
from torchlsq import extension as E, synth
import lsq_tools  # noqa: E402  (tools build of the library: `_ex` entry points, lsq_hip_debug_* knobs)
lsq_tools.activate()
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


for shape in ((8192, 4096), (32 * 2048, 4096), (2048, 8192), (65536, 1024), (12608, 768), (16384, 2048)):
    for dt in (torch.bfloat16, torch.float32):
        n = shape[0] * shape[1]
        x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt).view(shape)
        s = synth.uniform_like(shape[1], 3, 0.02, 0.05, device=dev); b = synth.normal_like(shape[1], 4, 0.0, 0.1, device=dev)
        q = (0, 127, 0, 255, True, 1.0, False, False, False)
        res = ["default %.1f" % timeit(lambda: E.hip_forward_per_channel(x, s, b, 1, *q))]
        for form, name in ((1, "reg"), (2, "ring")):
            for bpc in (2, 4, 8, 16):
                v = 4 | (3 << 8) | (bpc << 16) | (form << 12)
                res.append("%s/%d %.1f" % (name, bpc, timeit(lambda: E.hip_forward_per_channel(x, s, b, 1, *q, variant=v))))
        gb = n * x.element_size() * 2 / 1e3
        print("%-9s %-14s %s   (8 TB/s = %.1f us)" % (str(dt).replace("torch.", ""), shape, "  ".join(res), gb / 8e3), flush=True)
