#!/usr/bin/env python3
"""A/B on one box: the last-axis backward with row-group windows (default) against the 256-lane windows it replaced
(variant bit 11), GPU-side time (HIP graph of 20 launches, median of 7, three interleaved rounds)."""
import os
import sys

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
        gr.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


SHAPES = [((8192, 4096), 1), ((64, 197, 768), 2), ((64, 56, 56, 256), 3), ((65536, 1024), 1), ((32, 2048, 4096), 2), ((128, 768), 1)]
for dt in (torch.float32, torch.bfloat16):
    legacy = (4 | (3 << 8) | (2 << 16) | (1 << 11)) if dt == torch.float32 else (1 | (3 << 8) | (1 << 10) | (2 << 16) | (1 << 11))
    for shape, axis in SHAPES:
        n = 1
        for d in shape:
            n *= d
        x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt).view(shape)
        g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt).view(shape)
        C = shape[axis]
        s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev)
        b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        q = (0, 127, 0, 255, True, 1.0, False, False, False)
        res = {"new": [], "old": []}
        for _ in range(3):
            res["new"].append(timeit(lambda: E.hip_backward_per_channel(g, x, s, b, axis, *q)))
            res["old"].append(timeit(lambda: E.hip_backward_per_channel(g, x, s, b, axis, *q, variant=legacy)))
        esz = x.element_size()
        tn, to = min(res["new"]), min(res["old"])
        print("%-9s %-20s bwd row-group %8.2f us (%5.0f GB/s, %4.1f%% of 8 TB/s) | 256-lane windows %8.2f us | %+5.1f%%" %
              (str(dt).replace("torch.", ""), shape, tn, 3 * esz * n / tn / 1e3, 3 * esz * n / tn / 1e3 / 80, to, (tn / to - 1) * 100))
