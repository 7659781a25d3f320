#!/usr/bin/env python3
"""Is a tensor that the PREVIOUS kernel just wrote served from the Infinity Cache?  Per buffer set k (rotated through > 1 GiB, so
nothing survives from the previous visit): a producer writes x_k (ATen add of two other cold tensors), then the per-channel
forward reads x_k.  forward time = (graph of producer + forward) - (graph of producer alone), per launch; ring (bits 12-13 = 2)
vs register loops (1) vs default, next to the plain cold figure (no producer).  GPU-side us."""
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
REG, RING = 1 << 12, 2 << 12


def graph_time(body, reps):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        body(0)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for k in range(reps):
                body(k)
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


for shape, axis, dt in (((256, 2048, 7, 7), 1, torch.bfloat16), ((256, 2048, 7, 7), 1, torch.float32), ((64, 197, 768), 2, torch.float32),
                        ((64, 197, 768), 2, torch.bfloat16), ((32, 256, 56, 56), 1, torch.bfloat16), ((128, 512, 28, 28), 1, torch.bfloat16),
                        ((8192, 4096), 1, torch.bfloat16), ((64, 56, 56, 256), 3, torch.bfloat16)):
    n = 1
    for d in shape: n *= d
    esz = 4 if dt == torch.float32 else 2
    K = max(3, min(24, -(-(1100 << 20) // (n * esz * 3))))
    a_ = [synth.normal_like(n, 1 + k, 0.5, 1.0, device=dev, dtype=dt).view(shape) for k in range(K)]
    b_ = [synth.normal_like(n, 100 + k, 0.0, 1e-3, device=dev, dtype=dt).view(shape) for k in range(K)]
    xs = [torch.empty_like(a_[0]) for _ in range(K)]
    C = shape[axis]
    s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev); b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
    q = (-8, 7, -128, 127, True, 1.0, False, False, False)
    for k in range(K):
        torch.add(a_[k], b_[k], out=xs[k])
    t_prod = graph_time(lambda k: torch.add(a_[k % K], b_[k % K], out=xs[k % K]), K)
    res = ["producer alone %.1f" % t_prod]
    for name, v in (("default", 0), ("reg/16", 4 | (3 << 8) | (16 << 16) | REG), ("ring/4", 4 | (3 << 8) | (4 << 16) | RING),
                    ("ring/8", 4 | (3 << 8) | (8 << 16) | RING)):
        def both(k):
            torch.add(a_[k % K], b_[k % K], out=xs[k % K])
            E.hip_forward_per_channel(xs[k % K], s, b, axis, *q, variant=v)
        t_both = graph_time(both, K)
        t_cold = graph_time(lambda k: E.hip_forward_per_channel(xs[k % K], s, b, axis, *q, variant=v), K)
        res.append("%s: after producer %.1f, cold %.1f" % (name, t_both - t_prod, t_cold))
    print("%-9s %-18s x%d  fwd: %s" % (str(dt).replace("torch.", ""), shape, K, " | ".join(res)), flush=True)
    # backward: the gradient was just written by the producer, x is cold (saved by a forward long ago)
    import ctypes
    lib = E.library(); lib.lsq_hip_debug_set_ring_nt.argtypes = [ctypes.c_int]
    gsets = xs                                  # producer output = grad
    xcold = a_                                  # cold x (also one of the producer's inputs: read, not written, a rotation ago)
    res = []
    U = 1 if esz == 2 else 4
    pipe = (1 << 10) if esz == 2 else 0
    for name, v, knob in (("default", 0, 0), ("reg/4", U | (3 << 8) | (4 << 16) | REG | pipe, 0), ("reg/16", U | (3 << 8) | (16 << 16) | REG | pipe, 0),
                          ("ring/4", U | (3 << 8) | (4 << 16) | RING, 2), ("ring-nt/4", U | (3 << 8) | (4 << 16) | RING, 1),
                          ("ring/8", U | (3 << 8) | (8 << 16) | RING, 2), ("ring-nt/8", U | (3 << 8) | (8 << 16) | RING, 1)):
        lib.lsq_hip_debug_set_ring_nt(knob)
        E._WS_BYTES_PC.clear()
        def both(k):
            torch.add(a_[k % K], b_[k % K], out=gsets[k % K])
            E.hip_backward_per_channel(gsets[k % K], xcold[(k + K // 2) % K], s, b, axis, *q, variant=v)
        t_both = graph_time(both, K)
        t_cold = graph_time(lambda k: E.hip_backward_per_channel(gsets[k % K], xcold[(k + K // 2) % K], s, b, axis, *q, variant=v), K)
        res.append("%s: after producer %.1f, cold %.1f" % (name, t_both - t_prod, t_cold))
    lib.lsq_hip_debug_set_ring_nt(0)
    print("%-9s %-18s x%d  bwd: %s" % (str(dt).replace("torch.", ""), shape, K, " | ".join(res)), flush=True)
    del a_, b_, xs
