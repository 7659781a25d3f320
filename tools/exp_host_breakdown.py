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
