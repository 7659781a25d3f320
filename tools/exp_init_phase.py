#!/usr/bin/env python3
"""Cost of one initialisation-phase call of an observer-driven quantizer on the GPU (diagnostic): the one-launch tail
(lsq_hip_observer_update) against the reference sequence (observer forward, calculate_qparams, _set_weights): host
wall time per call, kernel launches and device synchronisations per call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch
from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
import torchlsq  # noqa: F401
from torchlsq.quantized import LSQFakeQuantizer

dev = torch.device("cuda:0")
for name, obs, kw, shape in (("per-tensor activation [32,256,28,28]", MovingAverageMinMaxObserver, {}, (32, 256, 28, 28)),
                             ("per-channel activation [32,256,28,28]", MovingAveragePerChannelMinMaxObserver,
                              dict(qscheme=torch.per_channel_affine), (32, 256, 28, 28)),
                             ("per-tensor activation [8,64,14,14] (small)", MovingAverageMinMaxObserver, {}, (8, 64, 14, 14))):
    x = torch.rand(shape, device=dev) * 3 - 1
    for fused in (True, False):
        m = LSQFakeQuantizer(obs, "activation", init_batches=10 ** 6, **kw).to(dev)
        m.fuse_observer_tail = fused
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e6
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
            for _ in range(10):
                m(x)
            torch.cuda.synchronize()
        ev = prof.events()
        launches = sum(1 for e in ev if e.name in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel", "cudaLaunchKernel"))
        syncs = sum(1 for e in ev if "Synchronize" in e.name or e.name in ("hipMemcpyWithStream", "hipMemcpyAsync", "cudaMemcpyAsync"))
        print("%-45s %-28s %7.1f us per call, %4.1f kernel launches, %4.1f syncs/copies per call" %
              (name, "one-launch tail" if fused else "reference sequence", dt, launches / 10.0, (syncs - 1) / 10.0), flush=True)
