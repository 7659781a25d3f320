#!/usr/bin/env python3
"""Per-channel activation quantizer (window mode mostly) on typical activation shapes: GPU-side forward / backward time (HIP graph), diagnostic."""
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


SHAPES = [((64, 3, 224, 224), 1), ((64, 64, 112, 112), 1), ((32, 256, 56, 56), 1), ((128, 512, 28, 28), 1), ((256, 2048, 7, 7), 1), ((128, 768), 1), ((8192, 4096), 1), ((65536, 1024), 1), ((64, 197, 768), 2), ((32, 2048, 4096), 2), ((64, 56, 56, 256), 3), ((1, 3, 2000, 2500), 1), ((4, 8, 1048576), 1)]
for dt in (torch.float32, torch.bfloat16):
    for shape, axis in SHAPES:
        n = 1
        for d in shape: n *= d
        x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt).view(shape)
        g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt).view(shape)
        C = shape[axis]
        s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev)
        b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        q = (0, 127, 0, 255)
        tf = timeit(lambda: ops.lsq_forward_per_channel(x, s, b, axis, *q, True, 1.0, False, False, False))
        tb = timeit(lambda: ops.lsq_backward_per_channel(g, x, s, b, axis, *q, True, 1.0, False, False, False))
        esz = x.element_size()
        print("%-9s %-22s axis %d n=%10d  fwd %8.2f us %6.0f GB/s | bwd %8.2f us %6.0f GB/s | fwd+bwd %6.1f GElem/s  %4.1f%% of 8 TB/s" %
              (str(dt).replace("torch.", ""), shape, axis, n, tf, 2 * esz * n / tf / 1e3, tb, 3 * esz * n / tb / 1e3, n / (tf + tb) / 1e3,
               5 * esz * n / (tf + tb) / 1e3 / 80))
