#!/usr/bin/env python3
"""One-pass min/max kernel vs torch.aminmax / the stock per-channel observer path (HIP-graph timing, diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torchlsq
from torchlsq import synth
dev = torch.device("cuda:0")
from torchlsq import extension as E
import lsq_tools  # noqa: E402  (tools build of the library: `_ex` entry points, lsq_hip_debug_* knobs)
lsq_tools.activate()
import ctypes
E._LIB.lsq_hip_debug_set_observe_wg_per_cu.argtypes = [ctypes.c_int]
WG = int(sys.argv[1]) if len(sys.argv) > 1 else 0
E._LIB.lsq_hip_debug_set_observe_wg_per_cu(WG)
print('wg/CU override:', WG)
ops = torch.ops.torchlsq

def timeit(fn, reps=10):
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

for name, shape, axis, dt in (("cfg2 per-tensor fp32", (128, 512, 56, 56), None, torch.float32),
                              ("cfg4s per-tensor fp32", (128, 1024, 14, 14), None, torch.float32),
                              ("cfg2 shape per-channel axis1 fp32", (128, 512, 56, 56), 1, torch.float32),
                              ("cfg5 per-channel axis1 fp32", (256, 2048, 7, 7), 1, torch.float32),
                              ("cfg5 per-channel axis1 bf16", (256, 2048, 7, 7), 1, torch.bfloat16),
                              ("cfg3 weights axis0 fp32", (512, 512, 3, 3), 0, torch.float32)):
    n = 1
    for d in shape: n *= d
    x = synth.normal_like(n, 5, 0.0, 1.0, device=dev, dtype=dt).view(shape)
    nb = n * x.element_size()
    if axis is None:
        t_new = timeit(lambda: ops.lsq_minmax_per_tensor(x))
        t_old = timeit(lambda: torch.aminmax(x))
    else:
        t_new = timeit(lambda: ops.lsq_minmax_per_channel(x, axis))
        def stock():
            order = list(range(x.dim())); order[axis] = 0; order[0] = axis
            y = torch.flatten(x.permute(order).to(torch.float32), start_dim=1)
            return torch.aminmax(y, dim=1)
        t_old = timeit(stock)
    print("%-36s one-pass kernel %8.2f us = %6.0f GB/s (%.1f%% of 8 TB/s) | torch path %8.2f us (%.1fx)" %
          (name, t_new, nb / t_new / 1e3, nb / t_new / 1e3 / 80, t_old, t_old / t_new))

# mean / unbiased std (3-sigma weight initialisation) against torch.mean + torch.std
for name, shape, axis, dt in (("cfg3 weights axis0 fp32", (512, 512, 3, 3), 0, torch.float32),
                              ("cfg3 weights per-tensor fp32", (512, 512, 3, 3), None, torch.float32),
                              ("linear weight [4096,11008] axis0 bf16", (4096, 11008), 0, torch.bfloat16),
                              ("cfg2 per-tensor fp32", (128, 512, 56, 56), None, torch.float32),
                              ("cfg5 per-channel axis1 fp32", (256, 2048, 7, 7), 1, torch.float32)):
    n = 1
    for d in shape: n *= d
    x = synth.normal_like(n, 5, 0.0, 1.0, device=dev, dtype=dt).view(shape)
    nb = n * x.element_size()
    if axis is None:
        t_new = timeit(lambda: ops.lsq_meanstd_per_tensor(x))
        t_old = timeit(lambda: (x.mean(), x.std()))
    else:
        dims = [d for d in range(x.dim()) if d != axis]
        t_new = timeit(lambda: ops.lsq_meanstd_per_channel(x, axis))
        t_old = timeit(lambda: (torch.mean(x, dims), torch.std(x, dims)))
    print("meanstd %-38s one-pass kernel %8.2f us = %6.0f GB/s (%.1f%% of 8 TB/s) | torch mean+std %8.2f us (%.1fx)" %
          (name, t_new, nb / t_new / 1e3, nb / t_new / 1e3 / 80, t_old, t_old / t_new))
