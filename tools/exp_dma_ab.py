#!/usr/bin/env python3
"""A/B on one box: window-mode per-channel backward with the LDS-DMA ring (variant bits 12-13 = 2) against the register
loops (= 1), several workgroups-per-CU settings each; first checks that both give the same bits."""
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


def code(dt, bpc, dma):
    base = (4 | (3 << 8)) if dt == torch.float32 else (1 | (3 << 8) | (1 << 10))
    return base | (bpc << 16) | (dma << 12)


def bits(t):
    t = t.detach().contiguous()
    return t.view(torch.int16 if t.element_size() == 2 else torch.int32).cpu().numpy().tobytes()


SHAPES = [((256, 2048, 7, 7), 1), ((32, 256, 56, 56), 1), ((8192, 4096), 1), ((64, 197, 768), 2), ((64, 56, 56, 256), 3),
          ((64, 3, 224, 224), 1), ((65536, 1024), 1)]
modes = sys.argv[1:] or ["train"]
for mode in modes:
    ev = mode == "eval"
    for dt in (torch.bfloat16, torch.float32):
        for shape, axis in SHAPES:
            n = 1
            for d in shape:
                n *= d
            x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt).view(shape)
            g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt).view(shape)
            C = shape[axis]
            s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev)
            b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
            q = (0, 127, 0, 255, True, 1.0, False, ev, False)
            if mode == "fwd":
                yr = E.hip_forward_per_channel(x, s, b, axis, *q, variant=code(dt, 16, 1))
                yd = E.hip_forward_per_channel(x, s, b, axis, *q, variant=code(dt, 16, 2))
                torch.cuda.synchronize()
                row = []
                for bpc in (4, 8, 16):
                    t_reg = timeit(lambda: E.hip_forward_per_channel(x, s, b, axis, *q, variant=code(dt, bpc, 1)))
                    t_dma = timeit(lambda: E.hip_forward_per_channel(x, s, b, axis, *q, variant=code(dt, bpc, 2)))
                    row.append("%d/CU %6.1f|%6.1f" % (bpc, t_reg, t_dma))
                t_def = timeit(lambda: E.hip_forward_per_channel(x, s, b, axis, *q))
                esz = x.element_size()
                print("%-5s %-9s %-20s y %s | default %6.1f us (%4.1f%%) | reg|dma us: %s" %
                      (mode, str(dt).replace("torch.", ""), shape, "same" if bits(yr) == bits(yd) else "DIFFERENT", t_def,
                       2 * esz * n / t_def / 1e3 / 80, "  ".join(row)), flush=True)
                continue
            ref = E.hip_backward_per_channel(g, x, s, b, axis, *q, variant=code(dt, 4, 1))
            got = E.hip_backward_per_channel(g, x, s, b, axis, *q, variant=code(dt, 4, 2))
            torch.cuda.synchronize()
            same = bits(ref[0]) == bits(got[0])
            err = max(float(((ref[k] - got[k]).abs() / ref[k].abs().clamp_min(1e-30)).max()) for k in (1, 2)) if not ev else 0.0
            row = []
            for bpc in (4, 8, 16):
                t_reg = timeit(lambda: E.hip_backward_per_channel(g, x, s, b, axis, *q, variant=code(dt, bpc, 1)))
                t_dma = timeit(lambda: E.hip_backward_per_channel(g, x, s, b, axis, *q, variant=code(dt, bpc, 2)))
                row.append("%d/CU %6.1f|%6.1f" % (bpc, t_reg, t_dma))
            t_def = timeit(lambda: E.hip_backward_per_channel(g, x, s, b, axis, *q))
            esz = x.element_size()
            print("%-5s %-9s %-20s dx %s ds/db rel %.1e | default %6.1f us (%4.1f%%) | reg|dma us: %s" %
                  (mode, str(dt).replace("torch.", ""), shape, "same" if same else "DIFFERENT", err, t_def,
                   3 * esz * n / t_def / 1e3 / 80, "  ".join(row)), flush=True)
