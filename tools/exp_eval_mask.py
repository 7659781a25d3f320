#!/usr/bin/env python3
"""eval-mode (plain fake-quant) step: masked path (1-byte inside mask saved) vs the x-based eval backward (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch, torchlsq
from torchlsq import synth, extension as E
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
for name, dt in (("cfg2", torch.float32), ("cfg4", torch.float32), ("cfg2", torch.bfloat16)):
    shape = list(synth.CONFIGS[name]["shape"])
    if name == "cfg4": shape[0] //= 8
    x, g, scale, shift = synth.make_inputs(name, device=dev, dtype=dt, shape=shape)
    q = (0, 127, 0, 255); n = x.numel(); es = x.element_size()
    t_f = timeit(lambda: E.hip_forward_per_tensor(x, scale, shift, *q, True, 1.0, False, True, False))
    t_fm = timeit(lambda: E.hip_forward_per_tensor(x, scale, shift, *q, True, 1.0, False, True, False, want_mask=True))
    _, mask = E.hip_forward_per_tensor(x, scale, shift, *q, True, 1.0, False, True, False, want_mask=True)
    t_b = timeit(lambda: E.hip_backward_per_tensor(g, x, scale, shift, *q, True, 1.0, False, True, False))
    t_bm = timeit(lambda: E.hip_backward_from_mask(g, mask))
    print("%s %s %s: fwd %.1f us -> fwd+mask %.1f us (%.0f GB/s); eval bwd from x %.1f us -> from mask %.1f us (%.0f GB/s); step %.1f -> %.1f us (%.1f%% faster); saved for backward %d -> %d MB" %
          (name, shape, str(dt).replace("torch.", ""), t_f, t_fm, (2 * es + 1) * n / t_fm / 1e3, t_b, t_bm, (2 * es + 1) * n / t_bm / 1e3,
           t_f + t_b, t_fm + t_bm, 100 * (1 - (t_fm + t_bm) / (t_f + t_b)), es * n >> 20, n >> 20))
