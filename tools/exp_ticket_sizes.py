#!/usr/bin/env python3
"""Per-tensor backward in ONE launch (arrival ticket, the last workgroup folds the partials) against kernel + finalize launch,
eager launches through the C++ binding, forward + backward per step: wall clock per step (the host queues 300 steps, one
synchronisation at the end) and GPU time of the backward alone (HIP-graph replay).  Which sizes are host-bound enough for the
saved launch to pay?  Output: profiles/r03_ticket_sizes.txt."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torchlsq  # noqa: E402,F401
from torchlsq import extension as E, synth  # noqa: E402

dev = torch.device("cuda:0")
ops = torch.ops.torchlsq_native
q = (0, 127, 0, 255, True, 1.0, False, False, False)


def wall(x, g, s, b, steps=300):
    for _ in range(20):
        ops.lsq_forward_per_tensor(x, s, b, *q); ops.lsq_backward_per_tensor(g, x, s, b, *q)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(steps):
            ops.lsq_forward_per_tensor(x, s, b, *q)
            ops.lsq_backward_per_tensor(g, x, s, b, *q)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    return best


def gpu_bwd(x, g, s, b, reps=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ops.lsq_backward_per_tensor(g, x, s, b, *q)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps):
                ops.lsq_backward_per_tensor(g, x, s, b, *q)
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[3]


def main():
    print("# " + __doc__.replace("\n", "\n# "))
    s = torch.tensor([0.03], device=dev)
    b = torch.tensor([0.0], device=dev)
    for dtype in (torch.float32, torch.bfloat16):
        for log2 in range(16, 26):
            n = 1 << log2
            x = synth.normal_like(n, 1, 1.5, 1.0, dtype=dtype, device=dev)
            g = synth.normal_like(n, 2, 0.0, 1e-3, dtype=dtype, device=dev)
            row = "%-9s 2^%d elements" % (str(dtype).replace("torch.", ""), log2)
            res = {}
            for on in (False, True):
                E.set_single_launch_backward(on)
                res[on] = wall(x, g, s, b)
            E.set_single_launch_backward("auto")
            print("%s | fwd+bwd wall per step: two launches %7.2f us, one launch %7.2f us (%+5.1f %%) | backward GPU time in a graph (no ticket there) %6.2f us"
                  % (row, res[False], res[True], (res[True] / res[False] - 1) * 100, gpu_bwd(x, g, s, b)), flush=True)


if __name__ == "__main__":
    main()
