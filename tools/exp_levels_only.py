#!/usr/bin/env python3
"""The conversion-time pass: int8 levels alone (lsq_hip_forward_* with y == NULL, 5 B per fp32 element) against the forward
that also writes y (9 B) and the plain forward (8 B), cold inputs, GPU time per launch from a HIP-graph replay.
    python tools/exp_levels_only.py                 # table -> profiles/r04_levels_only.txt
    python tools/exp_levels_only.py one cfg2        # a few launches of the levels-only op (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torchlsq  # noqa: E402,F401
from torchlsq import synth  # noqa: E402

dev = torch.device("cuda:0")
ops = torch.ops.torchlsq


def calls(cfg, dtype):
    c = synth.CONFIGS[cfg]
    x, _, scale, shift = synth.make_inputs(cfg, device=dev, dtype=dtype)
    q = (c["qmin"], c["qmax"], c["tmin"], c["tmax"])
    bias = 0
    if c["per_channel"]:
        return x, (lambda t: ops.lsq_levels_per_channel(t, scale, shift, c["axis"], *q, bias),
                   lambda t: ops.lsq_quantize_per_channel(t, scale, shift, c["axis"], *q, 128 if c["qmax"] > 127 else 0),
                   lambda t: ops.lsq_forward_per_channel(t, scale, shift, c["axis"], *q, True, 1.0, not c["affine"], False, False))
    return x, (lambda t: ops.lsq_levels_per_tensor(t, scale, shift, *q, bias),
               lambda t: ops.lsq_quantize_per_tensor(t, scale, shift, *q, 128 if c["qmax"] > 127 else 0),
               lambda t: ops.lsq_forward_per_tensor(t, scale, shift, *q, True, 1.0, not c["affine"], False, False))


def gpu_time(fn, xs, reps):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for t in xs:
            fn(t)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for k in range(reps):
                fn(xs[k % len(xs)])
        gr.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "one":
        x, (lv, _, _) = calls(sys.argv[2], torch.float32)
        for _ in range(4):
            lv(x)
        torch.cuda.synchronize()
        return
    print("# tools/exp_levels_only.py on one MI355X: GPU time per launch (HIP-graph replay, inputs rotated through > 1 GB), us; TB/s at the op's algorithmic bytes")
    for cfg, dtype in (("cfg2", torch.float32), ("cfg5", torch.float32), ("cfg5", torch.bfloat16), ("cfg3", torch.float32), ("cfg1", torch.float32)):
        x, fns = calls(cfg, dtype)
        esz = x.element_size()
        K = max(1, min(16, -(-(1 << 30) // (x.numel() * esz))))
        xs = [x] + [x.clone() for _ in range(K - 1)]
        out = []
        for name, fn, bpe in (("levels only", fns[0], esz + 1), ("y + levels", fns[1], 2 * esz + 1), ("forward (y)", fns[2], 2 * esz)):
            t = gpu_time(fn, xs, max(2 * K, 8))
            out.append("%s %8.1f us %5.2f TB/s (%d B/el)" % (name, t, bpe * x.numel() / t / 1e6, bpe))
        print("%-5s %-8s %-18s %s" % (cfg, str(dtype).replace("torch.", ""), list(x.shape), " | ".join(out)), flush=True)
        del xs, x


if __name__ == "__main__":
    main()
