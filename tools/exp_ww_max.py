#!/usr/bin/env python3
"""Where should last-axis tensors leave the row-group windows for the 256-lane ones?  (policy: 2^27 elements)

For [rows, C] tensors from 2^26 to 2^29 elements the backward op is timed with the row groups kept up to 2^40 elements
(lsq_hip_debug_set_ww_max_log2(40)) and with the 256-lane windows from 2^20 on (…(20)), three interleaved rounds, cold inputs
(rotated through > 1 GB), HIP-graph replay.  Output: profiles/r03_ww_max_ab.txt."""
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
MB = 1 << 20


def time_bwd(shape, dtype, settings, knob="set_ww_max_log2", axis=1, fwd=False):
    setter = getattr(lib, "lsq_hip_debug_" + knob)
    n = 1
    for d in shape:
        n *= d
    esz = 2 if dtype == torch.bfloat16 else 4
    K = max(2, min(8, -(-(1100 * MB) // (2 * n * esz))))
    xs = [synth.normal_like(n, 10 + k, 0.5, 1.0, dtype=dtype, device=dev).view(shape) for k in range(K)]
    gs = [synth.normal_like(n, 50 + k, 0.0, 1e-3, dtype=dtype, device=dev).view(shape) for k in range(K)]
    s = synth.uniform_like(shape[axis], 3, 0.01, 0.05, device=dev)
    b = synth.normal_like(shape[axis], 4, 0.0, 0.1, device=dev)
    q = (0, 127, 0, 255, True, 1.0, False, False, False)
    if fwd:
        op = lambda k: E.hip_forward_per_channel(xs[k], s, b, axis, *q)
    else:
        op = lambda k: E.hip_backward_per_channel(gs[k], xs[(k + K // 2) % K], s, b, axis, *q)
    graphs = {}
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for name, v in settings:
            setter(v)
            for k in range(K):
                op(k)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for k in range(2 * K):
                    op(k % K)
            note = lsq_tools.last_launch()
            graphs[name] = (gr, "%s %dx%d of %d lanes%s" % (note["kind"], note["grid_x"], note["grid_y"], note["block"],
                                                        ", ring %d" % note["ring_depth"] if note["ring_depth"] else ""))
        setter(0)
        out = {name: [] for name, _ in settings}
        for _ in range(3):
            for name, _v in settings:
                gr = graphs[name][0]
                gr.replay()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); gr.replay(); e1.record(); e1.synchronize()
                    ts.append(e0.elapsed_time(e1) / (2 * K) * 1e3)
                out[name].append(sorted(ts)[2])
    res = {name: (min(v), graphs[name][1]) for name, v in out.items()}
    del xs, gs, graphs
    torch.cuda.empty_cache()
    return res


def main():
    print(__doc__.split("\n\n")[1].replace("\n", " "))
    for dtype in (torch.bfloat16, torch.float32):
        for C in (768, 1024, 2048, 4096, 8192):
            for log2 in (26, 27, 28, 29):
                rows = (1 << log2) // C
                r = time_bwd((rows, C), dtype, (("row-groups", 40), ("windows", 20)))
                (tg, kg), (tw, kw) = r["row-groups"], r["windows"]
                n = rows * C
                print("%-9s [%7d,%5d] 2^%d elements  row groups %8.1f us %5.2f ps/el (%s) | 256-lane windows %8.1f us %5.2f ps/el (%s) | row groups %+5.1f %%"
                      % (str(dtype).replace("torch.", ""), rows, C, log2, tg, tg * 1e6 / n, kg, tw, tw * 1e6 / n, kw, (tg / tw - 1) * 100), flush=True)


if __name__ == "__main__":
    main()
