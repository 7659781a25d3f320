#!/usr/bin/env python3
"""A/B of one policy knob of the tools library on EAGER launches (tickets available, unlike under HIP-graph capture): the
backward op back to back on rotated inputs, wall time per op (GPU-bound sizes), three interleaved rounds.
    python tools/exp_eager_ab.py set_ww_cb 1 16 bf16 12608x768 50432x768"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torchlsq  # noqa: E402,F401
from torchlsq import extension as E, synth  # noqa: E402
import lsq_tools  # noqa: E402

lsq_tools.activate()
dev = torch.device("cuda:0")


def main():
    knob, a, b, dt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt]
    print("# lsq_hip_debug_%s(%d) against (%d), %s, backward op, EAGER launches (tickets on), us per op by wall clock" % (knob, a, b, dt))
    for spec in sys.argv[5:]:
        dims, _, ax = spec.partition("@")
        shape = tuple(int(v) for v in dims.split("x"))
        axis = int(ax) if ax else len(shape) - 1
        n = 1
        for d in shape:
            n *= d
        esz = 2 if dtype == torch.bfloat16 else 4
        K = max(2, min(12, -(-(1100 << 20) // (2 * n * esz))))
        xs = [synth.normal_like(n, 10 + k, 0.5, 1.0, dtype=dtype, device=dev).view(shape) for k in range(K)]
        gs = [synth.normal_like(n, 50 + k, 0.0, 1e-3, dtype=dtype, device=dev).view(shape) for k in range(K)]
        s = synth.uniform_like(shape[axis], 3, 0.01, 0.05, device=dev)
        bb = synth.normal_like(shape[axis], 4, 0.0, 0.1, device=dev)
        q = (0, 127, 0, 255, True, 1.0, False, False, False)
        res = {a: [], b: []}
        notes = {}
        for rnd in range(3):
            for v in (a, b):
                lsq_tools.set_knob(knob, v)
                for k in range(40):
                    E.hip_backward_per_channel(gs[k % K], xs[(k + K // 2) % K], s, bb, axis, *q)
                notes[v] = lsq_tools.last_launch()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(400):
                    E.hip_backward_per_channel(gs[k % K], xs[(k + K // 2) % K], s, bb, axis, *q)
                torch.cuda.synchronize()
                res[v].append((time.perf_counter() - t0) / 400 * 1e6)
        lsq_tools.set_knob(knob, 0)
        fmt = lambda nt: "%s %dx%d of %d lanes" % (nt["kind"], nt["grid_x"], nt["grid_y"], nt["block"])
        ta, tb = min(res[a]), min(res[b])
        print("%-20s %d: %6.1f us (%s) | %d: %6.1f us (%s) | %+5.1f %%" % (spec, a, ta, fmt(notes[a]), b, tb, fmt(notes[b]), (ta / tb - 1) * 100), flush=True)
        del xs, gs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
