#!/usr/bin/env python3
"""Window-mode per-channel kernels on COLD buffers (x and grad each rotated through > 1 GiB so that the 256 MB Infinity Cache cannot
serve them): register loops vs LDS-DMA ring (plain / nt copies) x workgroups per CU.  GPU-side us per call (HIP graph)."""
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
REG, RING = 1 << 12, 2 << 12


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
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


SHAPES = (((256, 2048, 7, 7), 1), ((128, 512, 28, 28), 1), ((32, 256, 56, 56), 1), ((8192, 4096), 1), ((65536, 1024), 1),
          ((64, 56, 56, 256), 3), ((256, 197, 768), 2))
if len(sys.argv) > 1:
    SHAPES = SHAPES[int(sys.argv[1]):]
for shape, axis in SHAPES:
    for dt in (torch.float32, torch.bfloat16):
        n = 1
        for d in shape: n *= d
        esz = 4 if dt == torch.float32 else 2
        copies = max(2, min(24, -(-(1100 << 20) // (n * esz))))      # > 1 GiB of x alone: cold for the forward too
        xs = [synth.normal_like(n, 1 + k, 0.5, 1.0, device=dev, dtype=dt).view(shape) for k in range(copies)]
        gs = [synth.normal_like(n, 100 + k, 0.0, 1e-3, device=dev, dtype=dt).view(shape) for k in range(copies)]
        C = shape[axis]
        s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev); b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        q = (-8, 7, -128, 127, True, 1.0, False, False, False)
        reps = 2 * copies
        def run_b(v):
            E._WS_BYTES_PC.clear()
            return timeit([(lambda k=k: E.hip_backward_per_channel(gs[k], xs[k], s, b, axis, *q, variant=v)) for k in range(copies)], reps)
        def run_f(v):
            return timeit([(lambda k=k: E.hip_forward_per_channel(xs[k], s, b, axis, *q, variant=v)) for k in range(copies)], reps)
        lib.lsq_hip_debug_set_ring_nt(0)
        bw = ["default %.1f" % run_b(0)]; fw = ["default %.1f" % run_f(0)]
        lib.lsq_hip_debug_set_ring_nt(2)
        U = 1 if esz == 2 else 4
        for bpc in (2, 4, 8, 16):
            bw.append("reg/%d %.1f" % (bpc, run_b(U | (3 << 8) | (bpc << 16) | REG | ((1 << 10) if esz == 2 else 0))))
            fw.append("reg/%d %.1f" % (bpc, run_f(4 | (3 << 8) | (bpc << 16) | REG)))
        for knob, name in ((2, "ring"), (1, "ring-nt")):
            lib.lsq_hip_debug_set_ring_nt(knob)
            for bpc in (2, 4, 8, 16):
                bw.append("%s/%d %.1f" % (name, bpc, run_b(U | (3 << 8) | (bpc << 16) | RING)))
                fw.append("%s/%d %.1f" % (name, bpc, run_f(4 | (3 << 8) | (bpc << 16) | RING)))
        lib.lsq_hip_debug_set_ring_nt(0)
        print("%-9s %-18s x%d bwd (8 TB/s = %.1f): %s" % (str(dt).replace("torch.", ""), shape, copies, n * esz * 3 / 8e6, "  ".join(bw)), flush=True)
        print("%-9s %-18s x%d fwd (8 TB/s = %.1f): %s" % (str(dt).replace("torch.", ""), shape, copies, n * esz * 2 / 8e6, "  ".join(fw)), flush=True)
        del xs, gs
