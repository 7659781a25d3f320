#!/usr/bin/env python3
"""Per-tensor forward / backward across tensor sizes and workgroups-per-CU (production library; HIP-graph timing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torchlsq  # noqa: F401
from torchlsq import synth, extension as E
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
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


q = (0, 127, 0, 255)
for dt in (torch.float32, torch.bfloat16):
    for lg in (16, 18, 20, 21, 22, 23, 24, 25, 26):
        n = (1 << lg) + 4096
        x = synth.normal_like(n, 1, 1.5, 1.0, device=dev, dtype=dt)
        g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt)
        s = torch.tensor([0.03], device=dev); b = torch.tensor([0.0], device=dev)
        row = []
        for bpc in (0, 2, 4, 8, 16):
            vf = 0 if bpc == 0 else (4 | (1 << 8) | (1 << 9) | (bpc << 16))
            tf = timeit(lambda: E.hip_forward_per_tensor(x, s, b, *q, True, 1.0, False, False, False, variant=vf))
            tb = timeit(lambda: E.hip_backward_per_tensor(g, x, s, b, *q, True, 1.0, False, False, False, variant=vf))
            row.append((bpc, round(tf, 2), round(tb, 2)))
        esz = x.element_size()
        d = row[0]
        print("%-8s n=2^%d  default fwd %.2f us %5.0f GB/s  bwd %.2f us %5.0f GB/s | (wg/CU, fwd, bwd): %s" %
              (str(dt).replace("torch.", ""), lg, d[1], 2 * esz * n / d[1] / 1e3, d[2], 3 * esz * n / d[2] / 1e3, row[1:]))
