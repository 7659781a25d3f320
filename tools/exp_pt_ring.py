#!/usr/bin/env python3
"""[needs the experiment build: make -C lsqfakequantize-pytorch_amd/csrc EXPERIMENT=pt_ring]  Per-tensor kernels K1 / K2: register loops (shipped defaults) against the LDS-DMA ring at several depths and
workgroups per CU (variant: unroll field = depth selector, bits 12-13 = 2 for the ring, bits 16-23 workgroups per CU).
GPU-side us per call (HIP graph), results checked bit for bit (y, dx) / to fp64 rounding (d_scale, d_shift)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch
import torchlsq  # noqa: F401
from torchlsq import extension as E, synth
dev = torch.device("cuda:0")


def timeit(fn, reps):
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


RING = 2 << 12
for shape, dt in (((128, 512, 56, 56), torch.float32), ((128, 1024, 14, 14), torch.float32), ((128, 1024, 14, 14), torch.bfloat16),
                  ((4, 64, 56, 56), torch.float32), ((64, 512, 56, 56), torch.bfloat16)):
    n = 1
    for d in shape: n *= d
    reps = 5 if n > 1e8 else 20
    x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt).view(shape)
    g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt).view(shape)
    s = torch.tensor([0.03], device=dev); b = torch.tensor([0.1], device=dev)
    tail = (0, 127, 0, 255, True, 1.0, False, False, False)
    y0 = E.hip_forward_per_tensor(x, s, b, *tail)
    r0 = E.hip_backward_per_tensor(g, x, s, b, *tail)
    esz = x.element_size()
    fw = ["default %.1f" % timeit(lambda: E.hip_forward_per_tensor(x, s, b, *tail), reps)]
    bw = ["default %.1f" % timeit(lambda: E.hip_backward_per_tensor(g, x, s, b, *tail), reps)]
    for ntl, nname in ((1, "nt"), (0, "plain")):
        for depth_sel, dname in ((4, "d8"), (8, "d16")):
            for bpc in (2, 4, 16):
                v = depth_sel | (ntl << 8) | (1 << 9) | (bpc << 16) | RING
                y1 = E.hip_forward_per_tensor(x, s, b, *tail, variant=v)
                assert torch.equal(y1.view(torch.int16 if esz == 2 else torch.int32), y0.view(torch.int16 if esz == 2 else torch.int32))
                fw.append("%s-%s/%d %.1f" % (nname, dname, bpc, timeit(lambda: E.hip_forward_per_tensor(x, s, b, *tail, variant=v), reps)))
        for depth_sel, dname in ((4, "d4"), (8, "d8")):
            for bpc in (1, 2, 4):
                v = depth_sel | (ntl << 8) | (1 << 9) | (bpc << 16) | RING
                r1 = E.hip_backward_per_tensor(g, x, s, b, *tail, variant=v)
                assert torch.equal(r1[0].view(torch.int16 if esz == 2 else torch.int32), r0[0].view(torch.int16 if esz == 2 else torch.int32))
                assert torch.allclose(r1[1].double(), r0[1].double(), rtol=1e-6, atol=1e-12) and torch.allclose(r1[2].double(), r0[2].double(), rtol=1e-6, atol=1e-12)
                bw.append("%s-%s/%d %.1f" % (nname, dname, bpc, timeit(lambda: E.hip_backward_per_tensor(g, x, s, b, *tail, variant=v), reps)))
    print("%-9s %-20s fwd (8 TB/s = %.1f us): %s" % (str(dt).replace("torch.", ""), shape, n * esz * 2 / 8e6, "  ".join(fw)), flush=True)
    print("%-9s %-20s bwd (8 TB/s = %.1f us): %s" % (str(dt).replace("torch.", ""), shape, n * esz * 3 / 8e6, "  ".join(bw)), flush=True)
