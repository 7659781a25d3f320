#!/usr/bin/env python3
"""What do the vendor / framework kernels reach on the same traffic shapes at config-2 size?  (diagnostic)
  1R:1W  y.copy_(x) (ATen copy kernel), hipMemcpyAsync device-to-device
  2R:1W  torch.add(g, x, out=dx) (ATen vectorised elementwise kernel)
  2R:0W  torch.dot-like read-only reduction: (g * x).sum() is 2R:1W+..., so use torch.linalg.vecdot? -> skipped
next to this build's forward (1R:1W + arithmetic) and backward (2R:1W + arithmetic + reduction)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch
import torchlsq  # noqa: F401
from torchlsq import synth
dev = torch.device("cuda:0")
x, g, scale, shift = synth.make_inputs("cfg2", device=dev, dtype=torch.float32)
n = x.numel()
y = torch.empty_like(x); dx = torch.empty_like(x)
ops = torch.ops.torchlsq
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]


def timeit(fn, reps=10):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(st)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn(st)
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


rows = [
    ("1R:1W  ATen  y.copy_(x)", 8, lambda st: y.copy_(x)),
    ("1R:1W  hipMemcpyAsync device-to-device", 8, lambda st: hip.hipMemcpyAsync(y.data_ptr(), x.data_ptr(), 4 * n, 3, st.cuda_stream)),
    ("1R:1W  this build: lsq_forward_per_tensor", 8, lambda st: ops.lsq_forward_per_tensor(x, scale, shift, 0, 127, 0, 255, True, 1.0, False, False, False)),
    ("2R:1W  ATen  torch.add(g, x, out=dx)", 12, lambda st: torch.add(g, x, out=dx)),
    ("2R:1W  this build: lsq_backward_per_tensor (+ reduction, + finalize launch)", 12,
     lambda st: ops.lsq_backward_per_tensor(g, x, scale, shift, 0, 127, 0, 255, True, 1.0, False, False, False)),
]
for name, bpe, fn in rows:
    t = timeit(fn)
    print("%-78s %8.1f us  %6.0f GB/s  (%.1f%% of 8 TB/s)" % (name, t, bpe * n / t / 1e3, bpe * n / t / 1e3 / 80))
