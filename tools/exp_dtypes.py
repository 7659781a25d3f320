#!/usr/bin/env python3
"""per-tensor fwd/bwd by storage type at BASELINE config 2 size (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, torchlsq
from torchlsq import synth, extension as E
import lsq_tools  # noqa: E402  (tools build of the library: `_ex` entry points, lsq_hip_debug_* knobs)
lsq_tools.activate()
dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(); gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps): fn()
        gr.replay(); torch.cuda.synchronize(); ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[3]
for dt in (torch.float32, torch.bfloat16, torch.float16, torch.float64):
    x, g, scale, shift = synth.make_inputs("cfg2", device=dev, dtype=dt)
    q = (0, 127, 0, 255); n = x.numel(); es = x.element_size()
    for bpc_f, bpc_b in ((0, 0), (8, 2), (16, 4), (16, 8)):
        vf = 0 if not bpc_f else (4 | (3 << 8) | (bpc_f << 16)); vb = 0 if not bpc_b else (4 | (3 << 8) | (bpc_b << 16))
        tf = timeit(lambda: E.hip_forward_per_tensor(x, scale, shift, *q, True, 1.0, False, False, False, variant=vf))
        tb = timeit(lambda: E.hip_backward_per_tensor(g, x, scale, shift, *q, True, 1.0, False, False, False, variant=vb))
        print("%-9s wg/CU f%-2d b%-2d fwd %7.1f us %5.0f GB/s  bwd %7.1f us %5.0f GB/s  -> %6.1f GElem/s" %
              (str(dt).replace("torch.", ""), bpc_f, bpc_b, tf, 2 * es * n / tf / 1e3, tb, 3 * es * n / tb / 1e3, n / (tf + tb) / 1e3))
