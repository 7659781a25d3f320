#!/usr/bin/env python3
"""Weight tensors whose channel rows are shorter than one workgroup's span (256 lanes x 16 bytes): window kernels (policy,
lsq_hip_debug_set_seg_min_div(1)) against the segment walk -- one workgroup per channel, d_scale / d_shift finished in the
kernel, no finalize launch -- for rows of at least 1/4 and 1/8 of the span.  Forward and backward op, GPU time (HIP-graph
replay over 8 rotated tensors, best of three interleaved rounds).  Output: profiles/r03_seg_min_ab.txt."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torchlsq  # noqa: E402,F401
from torchlsq import extension as E, synth  # noqa: E402
import lsq_tools  # noqa: E402

lib = lsq_tools.activate()
dev = torch.device("cuda:0")
K = 8
SHAPES = [(2304, 768), (768, 768), (3072, 768), (768, 3072), (1000, 512), (4096, 1024), (64, 64, 3, 3), (128, 64, 3, 3),
          (128, 128, 3, 3), (512, 256, 1, 1), (256, 128, 1, 1), (512, 512, 3, 3)]
SETTINGS = (("policy", 1), ("1/4", 4), ("1/8", 8))


def graphs_for(shape, dtype):
    n = 1
    for d in shape:
        n *= d
    xs = [synth.normal_like(n, 10 + k, 0.0, 0.05, dtype=dtype, device=dev).view(shape) for k in range(K)]
    gs = [synth.normal_like(n, 50 + k, 0.0, 1e-3, dtype=dtype, device=dev).view(shape) for k in range(K)]
    s = synth.uniform_like(shape[0], 3, 5e-4, 2.5e-3, device=dev)
    b = synth.normal_like(shape[0], 4, 0.0, 1e-3, device=dev)
    q = (-128, 127, -128, 127, True, 1.0, True, False, False)
    out = {}
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for name, v in SETTINGS:
            lib.lsq_hip_debug_set_seg_min_div(v)
            for which, fn in (("fwd", lambda k: E.hip_forward_per_channel(xs[k], s, b, 0, *q)),
                              ("bwd", lambda k: E.hip_backward_per_channel(gs[k], xs[k], s, b, 0, *q))):
                fn(0)
                kind = lsq_tools.last_launch()["kind"]
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    for k in range(2 * K):
                        fn(k % K)
                out[(name, which)] = (gr, kind)
        lib.lsq_hip_debug_set_seg_min_div(0)
        res = {key: [] for key in out}
        for _ in range(3):
            for key, (gr, _) in out.items():
                gr.replay()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); gr.replay(); e1.record(); e1.synchronize()
                    ts.append(e0.elapsed_time(e1) / (2 * K) * 1e3)
                res[key].append(sorted(ts)[2])
    return {key: (min(v), out[key][1]) for key, v in res.items()}


BIG = [(8192, 768), (16384, 768), (50257, 768), (8192, 512), (32768, 512), (131072, 512), (8192, 256), (32768, 256),
       (131072, 256), (8192, 128), (32768, 128), (131072, 128)]


def main():
    shapes = BIG if "--big" in sys.argv else SHAPES
    print("# " + __doc__.split("\n\n")[0].replace("\n", "\n# "))
    if "--big" in sys.argv:
        print("# --big: many channels -- where does one workgroup per (short) channel row stop paying?")
    for dtype in (torch.float32, torch.bfloat16):
        for shape in shapes:
            r = graphs_for(shape, dtype)
            inner = 1
            for d in shape[1:]:
                inner *= d
            row = "%-9s %-18s row %5d" % (str(dtype).replace("torch.", ""), shape, inner)
            for which in ("fwd", "bwd"):
                base = r[("policy", which)][0]
                row += " | %s" % which
                for name, _ in SETTINGS:
                    t, kind = r[(name, which)]
                    row += "  %s %6.2f us (%s%s)" % (name, t, kind[:3], "" if name == "policy" else ", %+.0f %%" % ((t / base - 1) * 100))
            print(row, flush=True)


if __name__ == "__main__":
    main()
