#!/usr/bin/env python3
"""LDS-DMA ring copies with and without the streaming hint (global_load_lds_dwordx4 ... nt), window-mode per-channel
kernels, default launch policy otherwise.  GPU-side us per call (HIP graph; buffers rotated through > 256 MB so that the
Infinity Cache does not serve them)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torchlsq  # noqa: F401
from torchlsq import extension as E, synth
import lsq_tools  # noqa: E402  (tools build of the library: `_ex` entry points, lsq_hip_debug_* knobs)
lsq_tools.activate()
lib = E.library()
lib.lsq_hip_debug_set_ring_nt.argtypes = [ctypes.c_int]
dev = torch.device("cuda:0")


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


SHAPES = (((256, 2048, 7, 7), 1), ((128, 512, 28, 28), 1), ((64, 64, 112, 112), 1), ((8192, 4096), 1), ((65536, 1024), 1),
          ((64, 56, 56, 256), 3), ((32, 2048, 4096), 2), ((512, 2048, 7, 7), 1), ((256, 197, 768), 2))
for shape, axis in SHAPES:
    for dt in (torch.float32, torch.bfloat16):
        n = 1
        for d in shape: n *= d
        esz = 4 if dt == torch.float32 else 2
        copies = max(1, min(6, (600 << 20) // (n * esz * 3)))
        xs = [synth.normal_like(n, 1 + k, 0.5, 1.0, device=dev, dtype=dt).view(shape) for k in range(copies)]
        gs = [synth.normal_like(n, 100 + k, 0.0, 1e-3, device=dev, dtype=dt).view(shape) for k in range(copies)]
        C = shape[axis]
        s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev); b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        q = (-8, 7, -128, 127, True, 1.0, False, False, False)
        reps = 4 * copies if n > 1e8 else 6 * copies
        res = []
        for knob in (2, 1, 2, 1):
            lib.lsq_hip_debug_set_ring_nt(knob)
            E._WS_BYTES_PC.clear()
            tf = timeit([(lambda k=k: E.hip_forward_per_channel(xs[k], s, b, axis, *q)) for k in range(copies)], reps)
            tb = timeit([(lambda k=k: E.hip_backward_per_channel(gs[k], xs[k], s, b, axis, *q)) for k in range(copies)], reps)
            res.append("%s fwd %.1f bwd %.1f" % ("nt" if knob == 1 else "plain", tf, tb))
        lib.lsq_hip_debug_set_ring_nt(0)
        print("%-9s %-20s x%d  %s" % (str(dt).replace("torch.", ""), shape, copies, " | ".join(res)), flush=True)
        del xs, gs
