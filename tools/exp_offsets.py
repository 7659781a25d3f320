#!/usr/bin/env python3
"""Does the relative placement of the three streams (grad, x, dx) in HBM matter? (diagnostic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch, torchlsq
from torchlsq import synth
dev = torch.device("cuda:0")
ops = torch.ops.torchlsq
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
n = 128 * 512 * 56 * 56
x0, g0, scale, shift = synth.make_inputs("cfg2", device=dev, dtype=torch.float32)
pad = 1 << 22
bufx = torch.empty(n + pad, device=dev); bufg = torch.empty(n + pad, device=dev)
q = (0, 127, 0, 255)
for offx, offg in ((0, 0), (0, 1024), (0, 4096 + 64), (0, 65536 + 1024), (1024, 2048 + 256 * 1024), (0, 1 << 20), (4, 8)):
    x = bufx[offx:offx + n]; g = bufg[offg:offg + n]
    x.copy_(x0.view(-1)); g.copy_(g0.view(-1))
    tb = timeit(lambda: ops.lsq_backward_per_tensor(g, x, scale, shift, *q, True, 1.0, False, False, False))
    tf = timeit(lambda: ops.lsq_forward_per_tensor(x, scale, shift, *q, True, 1.0, False, False, False))
    print("x +%-7d g +%-8d elements: fwd %.1f us (%.0f GB/s)  bwd %.1f us (%.0f GB/s)%s" %
          (offx, offg, tf, 8 * n / tf / 1e3, tb, 12 * n / tb / 1e3, "   (unaligned -> element-wise path)" if (offx % 4 or offg % 4) else ""))
