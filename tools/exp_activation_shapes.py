#!/usr/bin/env python3
"""Per-channel activation quantizer (window mode mostly) on typical activation shapes: GPU-side forward / backward time (HIP
graph).  Two columns.  `cold`: x and grad rotate through buffer sets of more than 1.2 GB in total, so the 256 MB Infinity
Cache cannot serve them -- what the HBM roofline is about.  `fresh`: what a layer sees in a training step -- a producer
kernel (ATen add, also rotated) has just written the forward's x / the backward's gradient, the backward's x is cold; op
time = graph(producer + op) - graph(producer), +-1 us.  (Re-reading ONE set of buffers, the "hot" column of earlier
versions, favours whatever keeps lines cached and is not a situation a training step has.)  Shapes streaming more than
1 GB per step have the cold column only."""
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
        big = 3 * n * esz > 1e9
        copies = 2 if big else max(3, min(16, int(1.2e9 // (3 * n * esz)) + 1))
        xs = [synth.normal_like(n, 1 + k, 0.5, 1.0, device=dev, dtype=dt).view(shape) for k in range(copies)]
        gs = [synth.normal_like(n, 50 + k, 0.0, 1e-3, device=dev, dtype=dt).view(shape) for k in range(copies)]
        C = shape[axis]
        s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev)
        b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        q = (0, 127, 0, 255)
        K = copies
        reps = 8 if n > 1e8 else max(24, 2 * K)
        fwd = lambda k: ops.lsq_forward_per_channel(xs[k % K], s, b, axis, *q, True, 1.0, False, False, False)
        bwd = lambda k: ops.lsq_backward_per_channel(gs[k % K], xs[(k + K // 2) % K], s, b, axis, *q, True, 1.0, False, False, False)
        tf = timeit([(lambda k=k: fwd(k)) for k in range(K)], reps)
        tb = timeit([(lambda k=k: bwd(k)) for k in range(K)], reps)
        def col(name, tf, tb):
            return "%s: fwd %7.2f us %5.0f GB/s  bwd %7.2f us %5.0f GB/s  fwd+bwd %5.1f GElem/s %4.1f%% of 8 TB/s" % (
                name, tf, 2 * esz * n / tf / 1e3, tb, 3 * esz * n / tb / 1e3, n / (tf + tb) / 1e3, 5 * esz * n / (tf + tb) / 1e3 / 80)
        cols = [col("cold" if not big else "(> 1 GB per step)", tf, tb)]
        if not big:
            outs = [torch.empty_like(xs[0]) for _ in range(K)]          # what the producer writes
            prod = lambda k: torch.add(xs[k % K], gs[k % K], out=outs[k % K])
            tp = timeit([(lambda k=k: prod(k)) for k in range(K)], reps)
            tfp = timeit([(lambda k=k: (prod(k), ops.lsq_forward_per_channel(outs[k % K], s, b, axis, *q, True, 1.0, False, False, False)))
                          for k in range(K)], reps) - tp
            tbp = timeit([(lambda k=k: (prod(k), ops.lsq_backward_per_channel(outs[k % K], xs[(k + K // 2) % K], s, b, axis, *q, True, 1.0, False, False, False)))
                          for k in range(K)], reps) - tp
            cols.append(col("fresh", tfp, tbp))
            del outs
        print("%-9s %-20s axis %d n=%10d | %s" % (str(dt).replace("torch.", ""), shape, axis, n, " | ".join(cols)), flush=True)
        del xs, gs
