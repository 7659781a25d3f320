#!/usr/bin/env python3
"""Ablation: per-channel backward, train vs eval (dx only) vs sym, HIP-graph timing (diagnostic)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torchlsq
from torchlsq import synth, extension as E
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
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]

for name, dt in (("cfg5", torch.float32), ("cfg5", torch.bfloat16), ("cfg2pc", torch.float32)):
    if name == "cfg2pc":
        c = dict(synth.CONFIGS["cfg5"]); shape = (32, 512, 56, 56)
    else:
        c = synth.CONFIGS[name]; shape = c["shape"]
    x, g, scale, shift = synth.make_inputs(c, device=dev, dtype=dt, shape=shape)
    q = (c["qmin"], c["qmax"], c["tmin"], c["tmax"])
    n = x.numel(); esz = x.element_size()
    for bpc in (2, 4, 8):
        v = 4 | (1 << 8) | (1 << 9) | (bpc << 16)
        row = {}
        for label, kw in (("train", dict(sym=False, ev=False)), ("sym", dict(sym=True, ev=False)), ("eval", dict(sym=False, ev=True))):
            t = timeit(lambda: E.hip_backward_per_channel(g, x, scale, shift, c["axis"], *q, True, 1.0, kw["sym"], kw["ev"], False, variant=v))
            row[label] = (round(t, 2), round(3 * esz * n / t / 1e3, 0))
        t = timeit(lambda: E.hip_forward_per_channel(x, scale, shift, c["axis"], *q, True, 1.0, False, False, False, variant=v))
        row["fwd"] = (round(t, 2), round(2 * esz * n / t / 1e3, 0))
        print(name, str(dt).replace("torch.", ""), list(shape), "bpc", bpc, row)
