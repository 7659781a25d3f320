#!/usr/bin/env python3
"""A/B of one policy knob of the tools library, backward op (forward: prefix the shape with f:), cold inputs; a shape is
rows x C (quantized on the last axis) or d0xd1x...@axis:
    python tools/exp_knob_ab.py set_ww_big 1 2 bf16 65536x768 87381x768 256x2048x7x7@1 f:8192x4096 ...
(three interleaved rounds of HIP-graph replays, inputs rotated through > 1 GB; the launch each setting produced is printed)."""
import sys

import torch

from exp_ww_max import time_bwd


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
