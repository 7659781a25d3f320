#!/usr/bin/env python3
"""(32,2048,4096) and friends: forward / backward time per workgroups-per-CU setting, register loops vs LDS-DMA ring."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torchlsq  # noqa: F401
from torchlsq import extension as E, synth
import lsq_tools  # noqa: E402  (tools build of the library: `_ex` entry points, lsq_hip_debug_* knobs)
lsq_tools.activate()
from exp_dma_ab import timeit, code
dev = torch.device("cuda:0")
for shape, axis in (((32, 2048, 4096), 2), ((128, 512, 28, 28), 1), ((64, 64, 112, 112), 1), ((65536, 1024), 1)):
    for dt in (torch.float32, torch.bfloat16):
        n = 1
        for d in shape: n *= d
        x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt).view(shape)
        g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt).view(shape)
        C = shape[axis]
        s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev); b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        q = (0, 127, 0, 255, True, 1.0, False, False, False)
        rowf, rowb = [], []
        for bpc in (2, 4, 8, 16):
            tf1 = timeit(lambda: E.hip_forward_per_channel(x, s, b, axis, *q, variant=code(dt, bpc, 1)), reps=10)
            tf2 = timeit(lambda: E.hip_forward_per_channel(x, s, b, axis, *q, variant=code(dt, bpc, 2)), reps=10)
            tb1 = timeit(lambda: E.hip_backward_per_channel(g, x, s, b, axis, *q, variant=code(dt, bpc, 1)), reps=10)
            tb2 = timeit(lambda: E.hip_backward_per_channel(g, x, s, b, axis, *q, variant=code(dt, bpc, 2)), reps=10)
            rowf.append("%d/CU %.0f|%.0f" % (bpc, tf1, tf2)); rowb.append("%d/CU %.0f|%.0f" % (bpc, tb1, tb2))
        tfd = timeit(lambda: E.hip_forward_per_channel(x, s, b, axis, *q), reps=10)
        tbd = timeit(lambda: E.hip_backward_per_channel(g, x, s, b, axis, *q), reps=10)
        print("%-9s %-18s fwd default %.0f us; reg|dma: %s || bwd default %.0f us; reg|dma: %s" %
              (str(dt).replace("torch.", ""), shape, tfd, "  ".join(rowf), tbd, "  ".join(rowb)), flush=True)
