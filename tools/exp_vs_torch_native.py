#!/usr/bin/env python3
"""cfg2 / cfg5 forward+backward against what PyTorch-ROCm itself offers on the same GPU (diagnostic):
  * torch._fake_quantize_learnable_per_tensor_affine / _per_channel_affine  -- ATen's own LSQ-style learnable
    fake-quantize (gradients for scale and zero point), the closest stock equivalent of the reference op;
  * torch.fake_quantize_per_tensor_affine (cachemask forward + masked backward, no parameter gradients).
HIP-graph timing of forward + backward (autograd), same inputs as bench.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch
import torchlsq  # noqa: F401  (registers the ops)
from torchlsq import synth
from torchlsq.functional import lsq

dev = torch.device("cuda:0")


def time_step(step, reps=5):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                step()
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


for name, per_channel in (("cfg2", False), ("cfg5", True)):
    c = synth.CONFIGS[name]
    x, g, scale, shift = synth.make_inputs(name, device=dev, dtype=torch.float32)
    n = x.numel()
    xr = x.clone().requires_grad_(True)
    s = scale.clone().requires_grad_(True)
    b = shift.clone().requires_grad_(True)
    zp = torch.zeros_like(scale).requires_grad_(True)
    qmin, qmax = c["qmin"], c["qmax"]

    def ours():
        xr.grad = s.grad = b.grad = None
        y = lsq(xr, s, b, qmin, qmax, c["tmin"], c["tmax"], c.get("axis", 1), True, 1.0, c["affine"], per_channel)
        y.backward(g)

    def learnable():
        xr.grad = s.grad = zp.grad = None
        if per_channel:
            y = torch._fake_quantize_learnable_per_channel_affine(xr, s, zp, c["axis"], qmin, qmax, 1.0)
        else:
            y = torch._fake_quantize_learnable_per_tensor_affine(xr, s, zp, qmin, qmax, 1.0)
        y.backward(g)

    sd, zpi = scale.detach(), torch.zeros(scale.numel(), dtype=torch.int32, device=dev)

    def plain():
        xr.grad = None
        if per_channel:
            y = torch.fake_quantize_per_channel_affine(xr, sd, zpi, c["axis"], qmin, qmax)
        else:
            y = torch.fake_quantize_per_tensor_affine(xr, sd, zpi[:1], qmin, qmax)
        y.backward(g)

    t_ours, t_learn, t_plain = time_step(ours), time_step(learnable), time_step(plain)
    print("%s %s  this build %.1f us = %.1f GElem/s | torch learnable fake-quant %.1f us = %.1f GElem/s (%.1fx slower) | "
          "torch plain fake-quant (no d_scale/d_shift) %.1f us = %.1f GElem/s" %
          (name, list(x.shape), t_ours, n / t_ours / 1e3, t_learn, n / t_learn / 1e3, t_learn / t_ours, t_plain, n / t_plain / 1e3))
