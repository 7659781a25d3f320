#!/usr/bin/env python3
"""A/B of one policy knob of the tools library, backward op (forward: prefix the shape with f:), cold inputs; a shape is
rows x C (quantized on the last axis) or d0xd1x...@axis:
    python tools/exp_knob_ab.py set_ww_big 1 2 bf16 65536x768 87381x768 256x2048x7x7@1 f:8192x4096 ...
(three interleaved rounds of HIP-graph replays, inputs rotated through > 1 GB; the launch each setting produced is printed)."""
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
    K = max(2, min(int(os.environ.get("LSQ_AB_SETS", "8")), -(-(1100 * MB) // (2 * n * esz))))     # (LSQ_AB_SETS=400: small tensors cold too)
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
    knob, a, b, dt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[dt]
    print("# lsq_hip_debug_%s(%d) against (%d), %s, backward op, cold" % (knob, a, b, dt))
    for spec in sys.argv[5:]:
        fwd = spec.startswith("f:")
        body = spec[2:] if fwd else spec
        dims, _, ax = body.partition("@")
        shape = tuple(int(v) for v in dims.split("x"))
        axis = int(ax) if ax else len(shape) - 1
        r = time_bwd(shape, dtype, (("a", a), ("b", b)), knob, axis=axis, fwd=fwd)
        (ta, ka), (tb, kb) = r["a"], r["b"]
        n = 1
        for d in shape:
            n *= d
        print("%-3s %-22s axis %d %10d elements  %d: %8.1f us %5.2f ps/el (%s) | %d: %8.1f us %5.2f ps/el (%s) | %+5.1f %%"
              % ("fwd" if fwd else "bwd", "x".join(str(d) for d in shape), axis, n, a, ta, ta * 1e6 / n, ka, b, tb, tb * 1e6 / n, kb, (ta / tb - 1) * 100), flush=True)


if __name__ == "__main__":
    main()
