#!/usr/bin/env python3
"""Per-tensor kernels: workgroups per CU of the persistent grid against the tensor size (the defaults -- 16 forward, 2 backward --
were tuned on BASELINE config 2, 205 M elements); GPU time per op, HIP-graph replay, inputs from HBM (LSQ_AB_SETS-style rotation
through > 1 GB).  Output: profiles/r04_pt_grid.txt."""
import sys

import torch

import lsq_tools
from torchlsq import extension as E, synth

lib = lsq_tools.activate()
dev = torch.device("cuda:0")
MB = 1 << 20


def enc(unroll, bpc):
    return unroll | (1 << 8) | (1 << 9) | (bpc << 16)


def graph_time(fn, K):
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
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / (2 * K) * 1e3)
    return sorted(ts)[2]


def main():
    print("# tools/exp_pt_grid.py: per-tensor forward / backward op, us, by workgroups per CU (0 = the library's default launch); cold inputs")
    q = (0, 127, 0, 255, True, 1.0, False, False, False)
    for dt_name in ("f32", "bf16"):
        dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt_name]
        esz = 2 if dtype == torch.bfloat16 else 4
        for n in (1 << 18, 802816, 1 << 21, 1 << 23, 25690112, 1 << 26, 205520896):
            K = max(2, min(400, -(-(1100 * MB) // (2 * n * esz))))
            xs = [synth.normal_like(n, 10 + k, 1.5, 1.0, dtype=dtype, device=dev) for k in range(K)]
            gs = [synth.normal_like(n, 50 + k, 0.0, 1e-3, dtype=dtype, device=dev) for k in range(K)]
            s = torch.tensor([0.03], device=dev)
            b = torch.tensor([0.0], device=dev)
            row = []
            for fwd in (True, False):
                cells = []
                for bpc in (0, 1, 2, 4, 8, 16, 32):
                    v = enc(4, bpc) if bpc else 0
                    if fwd:
                        fn = lambda k: E.hip_forward_per_tensor(xs[k], s, b, *q, variant=v)
                    else:
                        fn = lambda k: E.hip_backward_per_tensor(gs[k], xs[(k + K // 2) % K], s, b, *q, variant=v)
                    cells.append("%s %6.1f" % ("dflt" if not bpc else "%2d/CU" % bpc, graph_time(fn, K)))
                row.append(("fwd: " if fwd else "bwd: ") + "  ".join(cells))
            print("%-4s %10d el  %s  |  %s" % (dt_name, n, row[0], row[1]), flush=True)
            del xs, gs
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
