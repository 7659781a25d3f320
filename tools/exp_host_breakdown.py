#!/usr/bin/env python3
"""Host time of one small op call through the C++ binding, split by what can be timed from outside: a no-argument op of the
same library (dispatcher round trip), torch.empty_like (one allocation), the forward and the backward op without waiting for
the GPU (the host's share), and the same through the default overload vs the overload packet."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch
import torchlsq  # noqa: F401
from torchlsq import synth
dev = torch.device("cuda:0")
ns = torch.ops.torchlsq_native
x, g, scale, shift = synth.make_inputs("cfg3", device=dev, dtype=torch.float32)
q = (-128, 127, -128, 127, True, 1.0, True, False, False)
N = 4000


def per_call(fn):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(N):
            fn()
        t = time.perf_counter() - t0
        torch.cuda.synchronize()
        best = min(best, t / N * 1e6)
    return best


fwd_d, bwd_d = ns.lsq_forward_per_channel.default, ns.lsq_backward_per_channel.default
rows = [("python loop + lambda call", lambda: None),
        ("no-argument op (dispatcher round trip)", lambda: ns._abi_version()),
        ("torch.empty_like(x)", lambda: torch.empty_like(x)),
        ("forward op, default overload", lambda: fwd_d(x, scale, shift, 0, *q)),
        ("forward op, overload packet", lambda: ns.lsq_forward_per_channel(x, scale, shift, 0, *q)),
        ("backward op, default overload", lambda: bwd_d(g, x, scale, shift, 0, *q)),
        ("torch.add(x, g) (ATen's own op: one allocation + one launch)", lambda: torch.add(x, g)),
        ("x.mul_(1.0) (no allocation, one launch)", lambda: x.mul_(1.0))]
print("# host microseconds per call, GPU not waited for (cfg3 tensors, %d calls, best of 5)" % N)
for name, fn in rows:
    print("%-64s %6.2f us" % (name, per_call(fn)))

# BASELINE config 1 -- the size real activation quantizers have (reference quantized/modules/observers.py:458-461 calls
# functional.lsq once per activation and step): the per-tensor ops, the functional entry point with autograd, the module
x1, g1, s1, b1 = synth.make_inputs("cfg1", device=dev, dtype=torch.float32)
q1 = (0, 127, 0, 255, True, 1.0, False, False, False)
f1, b1op = ns.lsq_forward_per_tensor.default, ns.lsq_backward_per_tensor.default
from torchlsq.functional import lsq
xr, sr, br = x1.clone().requires_grad_(True), s1.clone().requires_grad_(True), b1.clone().requires_grad_(True)


def step_functional():
    y = lsq(xr, sr, br, 0, 127, 0, 255)
    y.backward(g1)
    xr.grad = None; sr.grad = None; br.grad = None


rows1 = [("cfg1 forward op (per-tensor), default overload", lambda: f1(x1, s1, b1, *q1)),
         ("cfg1 backward op (per-tensor; one launch with its ticket)", lambda: b1op(g1, x1, s1, b1, *q1)),
         ("cfg1 forward + backward ops (the bench's step)", lambda: (f1(x1, s1, b1, *q1), b1op(g1, x1, s1, b1, *q1))),
         ("two ATen single-launch ops: torch.add(x, g); torch.mul(x, g)", lambda: (torch.add(x1, g1), torch.mul(x1, g1))),
         ("functional.lsq(...) forward + .backward(g) (C++ front op + autograd node)", step_functional)]
# the module in its steady state (observer off, LSQ learning on): what a QAT model pays per activation quantizer and step
from torchlsq.quantized import LSQFakeQuantizer
from torch.ao.quantization.observer import MovingAverageMinMaxObserver
mod = LSQFakeQuantizer(MovingAverageMinMaxObserver, "activation", init_batches=1).to(dev).train()
for _ in range(4):
    mod(x1)
xm = x1.clone().requires_grad_(True)


def step_module():
    y = mod(xm)
    y.backward(g1)
    xm.grad = None; mod.scale.grad = None; mod.shift.grad = None


rows1 += [("functional.lsq(...) forward only (requires_grad inputs)", lambda: lsq(xr, sr, br, 0, 127, 0, 255)),
          ("LSQFakeQuantizer(x) forward only, steady state", lambda: mod(xm)),
          ("LSQFakeQuantizer(x) forward + .backward(g)", step_module)]
print("# BASELINE config 1 [4,64,56,56] fp32 (0.8 M elements: ~3 us of GPU time per op -- the host is the bottleneck)")
for name, fn in rows1:
    print("%-78s %6.2f us" % (name, per_call(fn)))
