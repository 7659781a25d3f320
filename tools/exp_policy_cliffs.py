#!/usr/bin/env python3
"""Size sweep across every threshold of the per-channel launch policy: is any of them a cliff?

For each threshold (tests/test_policy_gpu.py pins the branch taken on either side) the tensor size is swept from 0.7x to 1.3x
of it -- the two points next to the threshold are the shapes just below / just above it -- and the forward and the backward op are
timed on the GPU (HIP-graph replay, inputs rotated through > 1 GB so that they come from HBM).  Reported per point: us per op
and ps per element; the step across the threshold, in ps per element, is flagged when it exceeds 10 %.
Output: profiles/r03_policy_cliffs.txt, r04_policy_cliffs.txt (another box, + the owner band's bound)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torchlsq  # noqa: E402,F401
from torchlsq import extension as E, synth  # noqa: E402

dev = torch.device("cuda:0")
MB = 1 << 20


def time_ops(shape, axis, dtype, q):
    n = 1
    for d in shape:
        n *= d
    esz = 2 if dtype == torch.bfloat16 else 4
    K = max(2, min(8, -(-(1100 * MB) // (2 * n * esz))))
    C = shape[axis]
    xs = [synth.normal_like(n, 10 + k, 0.5, 1.0, dtype=dtype, device=dev).view(shape) for k in range(K)]
    gs = [synth.normal_like(n, 50 + k, 0.0, 1e-3, dtype=dtype, device=dev).view(shape) for k in range(K)]
    s = synth.uniform_like(C, 3, 0.01, 0.05, device=dev)
    b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
    args = q + (True, 1.0, False, False, False)
    ops = torch.ops.torchlsq_native if E.native_lsq() is not None else torch.ops.torchlsq
    out = []
    for fn in (lambda k: ops.lsq_forward_per_channel(xs[k], s, b, axis, *args),
               lambda k: ops.lsq_backward_per_channel(gs[k], xs[(k + K // 2) % K], s, b, axis, *args)):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for k in range(K):
                fn(k)
            gr = torch.cuda.CUDAGraph()
            reps = 2 * K
            with torch.cuda.graph(gr, stream=st):
                for k in range(reps):
                    fn(k % K)
            gr.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) / reps * 1e3)
        out.append(sorted(ts)[len(ts) // 2])
    del xs, gs
    torch.cuda.empty_cache()
    return n, out[0], out[1]


def sweep(title, make_shape, threshold_elems, axis, dtype, q=(0, 127, 0, 255)):
    """make_shape(rows) -> shape; threshold in elements of the whole tensor"""
    per_row = 1
    for d in make_shape(1):
        per_row *= d
    lo = (threshold_elems - 1) // per_row
    hi = lo + 1 if (lo + 1) * per_row >= threshold_elems else lo + 2
    rows = [max(1, int(lo * f)) for f in (0.7, 0.85, 0.95)] + [lo, hi] + [int(hi * f) for f in (1.05, 1.15, 1.3)]
    print("## %s  (%s)" % (title, str(dtype).replace("torch.", "")))
    prev = None
    for r in rows:
        shape = make_shape(r)
        n, tf, tb = time_ops(shape, axis, dtype, q)
        pf, pb = tf * 1e6 / n, tb * 1e6 / n
        mark = ""
        if r == hi and prev is not None:
            df, db_ = pf / prev[0] - 1, pb / prev[1] - 1
            mark = "   <-- threshold: forward %+.1f %%, backward %+.1f %% per element%s" % (
                100 * df, 100 * db_, "   CLIFF" if max(abs(df), abs(db_)) > 0.10 else "")
        print("  %-22s %11d elements   forward %8.1f us %6.2f ps/el   backward %8.1f us %6.2f ps/el%s" % (shape, n, tf, pf, tb, pb, mark), flush=True)
        prev = (pf, pb)


def main():
    print("# tools/exp_policy_cliffs.py on one MI355X: GPU time per op around every size threshold of the per-channel launch policy")
    print("# (HIP-graph replay, inputs rotated through > 1 GB: cold).  The branch on either side is pinned by tests/test_policy_gpu.py.")
    f32, bf16 = torch.float32, torch.bfloat16
    last = lambda C: (lambda r: (r, C))
    for dt in (f32, bf16):
        sweep("2^21 elements: NO switch here any more -- rows per workgroup now follow a smooth rule (row-group windows, [rows,768]; round 2 had a cliff here)", last(768), 1 << 21, 1, dt)
    for dt in (f32, bf16):
        sweep("32 MB: streaming hint on the ring copies (256-lane windows, [rows,2048,7] axis 1)", lambda r: (r, 2048, 7),
              32 * MB // (4 if dt == f32 else 2) + 1, 1, dt, (-8, 7, -128, 127))
    for dt in (f32, bf16):
        sweep("2^23 elements: 16-bit storage, one 768-lane workgroup per CU from here; fp32: nothing ([rows,768])", last(768), 1 << 23, 1, dt)
    sweep("2^24 elements: fp32 row groups take the ring from here ([rows,1024])", last(1024), 1 << 24, 1, f32)
    sweep("5 * 2^24 elements: ... for 16-bit storage up to here ([rows,768])", last(768), 5 << 24, 1, bf16)
    sweep("160 MB: fp32 row groups leave the ring ([rows,768])", last(768), 160 * MB // 4 + 1, 1, f32)
    for dt in (f32, bf16):
        sweep("512 MB: row groups give way to 256-lane windows ([rows,2048])", last(2048), 512 * MB // (4 if dt == f32 else 2), 1, dt)
    # (round 4) owner windows: NCHW activations of at most 13 M (fp32) / 20 M (16-bit) elements take one launch without finalize
    sweep("13 * 2^20 elements: owner windows up to here, 256-lane windows + finalize above ([rows,2048,7,7] axis 1)",
          lambda r: (r, 2048, 7, 7), (13 << 20) + 1, 1, f32, (-8, 7, -128, 127))
    sweep("13 * 2^20 elements: owner windows up to here ([rows,512,14,14] axis 1)",
          lambda r: (r, 512, 14, 14), (13 << 20) + 1, 1, f32, (-8, 7, -128, 127))
    sweep("20 * 2^20 elements: owner windows up to here, 256-lane windows + finalize above ([rows,2048,7,7] axis 1)",
          lambda r: (r, 2048, 7, 7), (20 << 20) + 1, 1, bf16, (-8, 7, -128, 127))
    sweep("20 * 2^20 elements: owner windows up to here ([rows,512,14,14] axis 1)",
          lambda r: (r, 512, 14, 14), (20 << 20) + 1, 1, bf16, (-8, 7, -128, 127))
    sweep("5 * 2^20 elements: owner windows whose runs are under 512 bytes and not whole cache lines only up to here ([rows,2048,7] axis 1: 224-byte runs)",
          lambda r: (r, 2048, 7), (5 << 20) + 1, 1, f32, (-8, 7, -128, 127))
    sweep("2^24 elements: NO switch here any more -- the 16-bit last-axis forward rule was implied by its tiles-per-workgroup condition ([rows,4096])", last(4096), 1 << 24, 1, bf16)


if __name__ == "__main__":
    main()
