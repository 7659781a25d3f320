#!/usr/bin/env python3
"""The per-channel forward right AFTER a producer: x is written by an elementwise kernel (torch.add(a, b, out=x)) and quantized
by the next kernel, as in a network -- register loops (force_ring 1) against the LDS-DMA ring (force_ring 2) against the policy.
GPU time of the forward = (graph of K x [producer, forward]) - (graph of K x [producer]), K input sets rotated.
Output: profiles/r04_fwd_after_producer.txt."""
import sys

import torch

import lsq_tools
from torchlsq import extension as E, synth

lib = lsq_tools.activate()
dev = torch.device("cuda:0")
SHAPES = [((48, 2048, 8, 8), 1), ((128, 256, 14, 14), 1), ((32, 1024, 14, 14), 1), ((32, 56, 56, 64), 3), ((16, 512, 28, 28), 1), ((64, 2048, 7, 7), 1),
          ((16, 2048, 10, 10), 1), ((24, 2048, 7, 7), 1), ((128, 2048, 7, 7), 1), ((256, 2048, 7, 7), 1), ((64, 56, 56, 64), 3), ((32, 256, 56, 56), 1),
          ((12608, 768), 1), ((8192, 4096), 1)]


def graph_time(fn, K, reps=5):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for k in range(K):
            fn(k)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for k in range(2 * K):
                fn(k % K)
        gr.replay()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / (2 * K) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    print("# tools/exp_fwd_after_producer.py: per-channel forward right after the kernel that wrote its input; us of GPU time")
    for dt_name in sys.argv[1:] or ["bf16", "f32"]:
        dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt_name]
        for shape, axis in SHAPES:
            n = 1
            for d in shape:
                n *= d
            esz = 2 if dtype == torch.bfloat16 else 4
            K = max(2, min(8, -(-(600 << 20) // (3 * n * esz))))
            a = [synth.normal_like(n, 10 + k, 0.5, 1.0, dtype=dtype, device=dev).view(shape) for k in range(K)]
            b = [synth.normal_like(n, 30 + k, 0.0, 0.1, dtype=dtype, device=dev).view(shape) for k in range(K)]
            x = [torch.empty(shape, dtype=dtype, device=dev) for _ in range(K)]
            s = synth.uniform_like(shape[axis], 3, 0.01, 0.05, device=dev)
            sh = synth.normal_like(shape[axis], 4, 0.0, 0.1, device=dev)
            q = (0, 127, 0, 255, True, 1.0, False, False, False)
            prod = lambda k: torch.add(a[k], b[k], out=x[k])
            t_prod = graph_time(prod, K)
            out = {}
            for name, v in (("policy", 0), ("registers", 1), ("ring", 2)):
                lib.lsq_hip_debug_force_ring(v)
                both = lambda k: (prod(k), E.hip_forward_per_channel(x[k], s, sh, axis, *q))
                t = graph_time(both, K) - t_prod
                note = lsq_tools.last_launch()
                out[name] = (t, "%dx%d%s" % (note["grid_x"], note["grid_y"], ", ring %d" % note["ring_depth"] if note["ring_depth"] else ""))
            lib.lsq_hip_debug_force_ring(0)
            print("%-4s %-18s axis %d %9d el  producer %6.1f | forward after it: policy %6.1f [%s]  registers %6.1f [%s]  ring %6.1f [%s]   ring / registers %+5.1f %%" % (
                dt_name, "x".join(map(str, shape)), axis, n, t_prod, out["policy"][0], out["policy"][1], out["registers"][0], out["registers"][1],
                out["ring"][0], out["ring"][1], (out["ring"][0] / out["registers"][0] - 1) * 100), flush=True)
            del a, b, x
            torch.cuda.empty_cache()


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
