#!/usr/bin/env python3
"""Per-channel (axis 0) weight quantizer on typical weight shapes: GPU-side forward / backward time (HIP graph), diagnostic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch
import torchlsq  # noqa: F401
from torchlsq import synth
ops = torch.ops.torchlsq
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


SHAPES = [(64, 3, 7, 7), (256, 256, 3, 3), (512, 512, 3, 3), (1024, 1024, 3, 3), (2048, 512, 1, 1), (1000, 2048), (4096, 4096),
          (4096, 11008), (11008, 4096), (32000, 4096), (8, 1 << 22), (3, 5000000)]
for dt in (torch.float32, torch.bfloat16):
    for shape in SHAPES:
        n = 1
        for d in shape: n *= d
        x = synth.normal_like(n, 1, 0.0, 0.05, device=dev, dtype=dt).view(shape)
        g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt).view(shape)
        C = shape[0]
        s = synth.uniform_like(C, 3, 5e-4, 2.5e-3, device=dev)
        b = torch.zeros(C, device=dev)
        q = (-128, 127, -128, 127)
        tf = timeit(lambda: ops.lsq_forward_per_channel(x, s, b, 0, *q, True, 1.0, True, False, False))
        tb = timeit(lambda: ops.lsq_backward_per_channel(g, x, s, b, 0, *q, True, 1.0, True, False, False))
        esz = x.element_size()
        print("%-9s %-22s n=%10d  fwd %8.2f us %6.0f GB/s | bwd %8.2f us %6.0f GB/s | fwd+bwd %6.1f GElem/s  %4.1f%% of 8 TB/s" %
              (str(dt).replace("torch.", ""), shape, n, tf, 2 * esz * n / tf / 1e3, tb, 3 * esz * n / tb / 1e3, n / (tf + tb) / 1e3,
               5 * esz * n / (tf + tb) / 1e3 / 80))
