#!/usr/bin/env python3
"""[needs the experiment build: make -C lsqfakequantize-pytorch_amd/csrc EXPERIMENT=pt_ring]  Per-tensor backward / forward, fp32 and bf16: register loops (default) vs the LDS-DMA ring (4 stages, 1 or 2 workgroups
per CU, plain loads) over tensor sizes.  GPU-side us per call (HIP graph), buffers rotated when they would sit in the 256 MB
Infinity Cache."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch
import torchlsq  # noqa: F401
from torchlsq import extension as E, synth
dev = torch.device("cuda:0")
RING = 2 << 12


def timeit(fns, reps):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for f in fns: f()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for k in range(reps):
                fns[k % len(fns)]()
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


for dt in (torch.float32, torch.bfloat16):
    for n in (1 << 21, 3211264, 6422528, 12845056, 25690112, 51380224, 102760448, 205520896):
        esz = 4 if dt == torch.float32 else 2
        copies = max(1, min(8, (600 << 20) // (n * esz * 3)))       # rotate through > 256 MB of buffers
        xs = [synth.normal_like(n, 1 + k, 0.5, 1.0, device=dev, dtype=dt) for k in range(copies)]
        gs = [synth.normal_like(n, 100 + k, 0.0, 1e-3, device=dev, dtype=dt) for k in range(copies)]
        s = torch.tensor([0.03], device=dev); b = torch.tensor([0.1], device=dev)
        tail = (0, 127, 0, 255, True, 1.0, False, False, False)
        reps = 4 * copies if n > 5e7 else 24
        out = []
        for name, v in (("default", 0), ("ring4/1", 4 | (2 << 8) | (1 << 16) | RING), ("ring4/2", 4 | (2 << 8) | (2 << 16) | RING),
                        ("ring4/4", 4 | (2 << 8) | (4 << 16) | RING)):
            fns = [(lambda k=k: E.hip_backward_per_tensor(gs[k], xs[k], s, b, *tail, variant=v)) for k in range(copies)]
            out.append("%s %.1f" % (name, timeit(fns, reps)))
        fo = []
        for name, v in (("default", 0), ("ring8/2", 4 | (2 << 8) | (2 << 16) | RING), ("ring8/16", 4 | (2 << 8) | (16 << 16) | RING)):
            fns = [(lambda k=k: E.hip_forward_per_tensor(xs[k], s, b, *tail, variant=v)) for k in range(copies)]
            fo.append("%s %.1f" % (name, timeit(fns, reps)))
        print("%-9s n=%-10d (x%d buffers)  bwd: %s   | fwd: %s" % (str(dt).replace("torch.", ""), n, copies, "  ".join(out), "  ".join(fo)), flush=True)
        del xs, gs
