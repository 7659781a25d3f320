#!/usr/bin/env python3
"""How much of the per-channel kernels' time is per-channel handling?  The per-tensor ops (no channel table, no partial rows)
on the same tensors as the per-channel ones: GPU time per op, cold inputs (rotated through > 1 GB), HIP-graph replay; plus
ATen's copy / add on the same bytes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torchlsq  # noqa: E402,F401
from torchlsq import extension as E, synth  # noqa: E402

dev = torch.device("cuda:0")
MB = 1 << 20


def replay_time(fn, K):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for k in range(K):
            fn(k)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for k in range(2 * K):
                fn(k % K)
        gr.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / (2 * K) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    ops = torch.ops.torchlsq_native if E.native_lsq() is not None else torch.ops.torchlsq
    for shape, axis in (((256, 2048, 7, 7), 1), ((8192, 4096), 1), ((64, 197, 768), 2)):
        for dtype in (torch.bfloat16, torch.float32):
            n = 1
            for d in shape:
                n *= d
            esz = 2 if dtype == torch.bfloat16 else 4
            K = max(2, min(12, -(-(1100 * MB) // (3 * n * esz))))
            C = shape[axis]
            xs = [synth.normal_like(n, 10 + k, 0.5, 1.0, dtype=dtype, device=dev).view(shape) for k in range(K)]
            gs = [synth.normal_like(n, 50 + k, 0.0, 1e-3, dtype=dtype, device=dev).view(shape) for k in range(K)]
            outs = [torch.empty_like(xs[0]) for _ in range(K)]
            s = synth.uniform_like(C, 3, 0.01, 0.05, device=dev)
            b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
            s1, b1 = s[:1].clone(), b[:1].clone()
            q = (0, 127, 0, 255, True, 1.0, False, False, False)
            t = {
                "pc fwd": replay_time(lambda k: ops.lsq_forward_per_channel(xs[k], s, b, axis, *q), K),
                "pt fwd": replay_time(lambda k: ops.lsq_forward_per_tensor(xs[k], s1, b1, *q), K),
                "copy": replay_time(lambda k: outs[k].copy_(xs[k]), K),
                "pc bwd": replay_time(lambda k: ops.lsq_backward_per_channel(gs[k], xs[(k + K // 2) % K], s, b, axis, *q), K),
                "pt bwd": replay_time(lambda k: ops.lsq_backward_per_tensor(gs[k], xs[(k + K // 2) % K], s1, b1, *q), K),
                "add": replay_time(lambda k: torch.add(gs[k], xs[(k + K // 2) % K], out=outs[k]), K),
            }
            fb, bb = 2 * esz * n, 3 * esz * n
            print("%-9s %-18s | forward: per-channel %6.2f us (%4.0f GB/s)  per-tensor %6.2f (%4.0f)  aten copy %6.2f (%4.0f) | backward: per-channel %6.2f us (%4.0f GB/s)  per-tensor %6.2f (%4.0f)  aten add %6.2f (%4.0f)"
                  % (str(dtype).replace("torch.", ""), shape, t["pc fwd"], fb / t["pc fwd"] / 1e3, t["pt fwd"], fb / t["pt fwd"] / 1e3, t["copy"], fb / t["copy"] / 1e3,
                     t["pc bwd"], bb / t["pc bwd"] / 1e3, t["pt bwd"], bb / t["pt bwd"] / 1e3, t["add"], bb / t["add"] / 1e3), flush=True)
            del xs, gs, outs
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
