#!/usr/bin/env python3
"""Per-channel activation quantizer (window mode mostly) on typical activation shapes: GPU-side forward / backward time (HIP
graph).  Two columns: `hot` re-uses one set of buffers (the 256 MB Infinity Cache serves part or all of the reads of the
smaller shapes -- what a layer sees when its input was just produced), `cold` rotates through buffer sets of more than
1.2 GB in total (what the HBM roofline is about); shapes streaming more than 1 GB per step have one column."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch
import torchlsq  # noqa: F401
from torchlsq import synth
ops = torch.ops.torchlsq
dev = torch.device("cuda:0")


def timeit(fns, reps=24):
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


SHAPES = [((64, 3, 224, 224), 1), ((64, 64, 112, 112), 1), ((32, 256, 56, 56), 1), ((128, 512, 28, 28), 1), ((256, 2048, 7, 7), 1), ((128, 768), 1), ((8192, 4096), 1), ((65536, 1024), 1), ((64, 197, 768), 2), ((32, 2048, 4096), 2), ((64, 56, 56, 256), 3), ((1, 3, 2000, 2500), 1), ((4, 8, 1048576), 1)]
for dt in (torch.float32, torch.bfloat16):
    for shape, axis in SHAPES:
        n = 1
        for d in shape: n *= d
        esz = 4 if dt == torch.float32 else 2
        copies = 1 if 3 * n * esz > 1e9 else max(2, min(16, int(1.2e9 // (3 * n * esz)) + 1))
        xs = [synth.normal_like(n, 1 + k, 0.5, 1.0, device=dev, dtype=dt).view(shape) for k in range(copies)]
        gs = [synth.normal_like(n, 50 + k, 0.0, 1e-3, device=dev, dtype=dt).view(shape) for k in range(copies)]
        C = shape[axis]
        s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev)
        b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        q = (0, 127, 0, 255)
        cols = []
        for mode in (("hot", "cold") if copies > 1 else ("cold",)):
            ks = range(copies) if mode == "cold" else (0,)
            reps = 8 if n > 1e8 else max(24, 2 * copies)
            tf = timeit([(lambda k=k: ops.lsq_forward_per_channel(xs[k], s, b, axis, *q, True, 1.0, False, False, False)) for k in ks], reps)
            tb = timeit([(lambda k=k: ops.lsq_backward_per_channel(gs[k], xs[k], s, b, axis, *q, True, 1.0, False, False, False)) for k in ks], reps)
            cols.append("%s: fwd %7.2f us %5.0f GB/s  bwd %7.2f us %5.0f GB/s  fwd+bwd %5.1f GElem/s %4.1f%% of 8 TB/s" %
                        (mode if copies > 1 else "(> 1 GB per step)", tf, 2 * esz * n / tf / 1e3, tb, 3 * esz * n / tb / 1e3, n / (tf + tb) / 1e3,
                         5 * esz * n / (tf + tb) / 1e3 / 80))
        print("%-9s %-20s axis %d n=%10d | %s" % (str(dt).replace("torch.", ""), shape, axis, n, " | ".join(cols)), flush=True)
        del xs, gs
