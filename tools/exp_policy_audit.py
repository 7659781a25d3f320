#!/usr/bin/env python3
"""Audit of the per-channel backward's launch policy: for a spread of shapes (NCHW activations, token layouts, conv / linear
weights; seeded), time the op as the policy launches it and with each family-forcing knob of the tools build, and report where a
forced alternative beats the policy by more than 7 %.  backward op, cold inputs, HIP-graph replay.
Output: profiles/r04_policy_audit.txt."""
import sys

import numpy as np
import torch

from exp_knob_ab import time_bwd
import lsq_tools

ALTS = [("set_own", 1), ("set_own", 2), ("set_ww_big", 1), ("set_ww_big", 2), ("force_ring", 1), ("force_ring", 2), ("set_seg_min_div", 1)]
ALTS_FWD = [("force_ring", 1), ("force_ring", 2), ("set_fwd_direct", 1), ("set_fwd_direct", 2), ("set_fwd_direct", 4), ("set_seg_min_div", 1)]


def shapes(rng, count):
    out = []
    hw = [(7, 7), (14, 14), (28, 28), (56, 56), (4, 4), (8, 8), (3, 3), (5, 5), (10, 10), (16, 16), (32, 32)]
    while len(out) < count:
        kind = rng.integers(0, 4)
        if kind == 0:                                   # NCHW activation, axis 1
            n, c = int(rng.choice([8, 16, 24, 32, 48, 64, 96, 128, 256])), int(rng.choice([64, 128, 256, 512, 1024, 2048]))
            h, w = hw[rng.integers(0, len(hw))]
            s, ax = (n, c, h, w), 1
        elif kind == 1:                                 # tokens x features, last axis
            s, ax = (int(rng.choice([197 * 16, 197 * 64, 1024, 4096, 8192, 16384, 50000])), int(rng.choice([384, 768, 1024, 1280, 2048, 4096]))), 1
        elif kind == 2:                                 # conv / linear weight, axis 0
            c = int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096]))
            k = int(rng.choice([9 * 64, 9 * 128, 9 * 256, 9 * 512, 768, 1024, 3072, 4096, 49 * 3]))
            s, ax = (c, k), 0
        else:                                           # NHWC activation, last axis
            n, c = int(rng.choice([8, 16, 32, 64])), int(rng.choice([64, 128, 256, 512]))
            h, w = hw[rng.integers(0, 4)]
            s, ax = (n, h, w, c), 3
        el = int(np.prod(s))
        if 200_000 <= el <= 60_000_000 and (s, ax) not in out:
            out.append((s, ax))
    return out


def main():
    fwd = "--forward" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    count = int(args[0]) if args else 40
    rng = np.random.default_rng(2024)
    print("# tools/exp_policy_audit.py: %s op, us, cold inputs; policy = the production launch; every alternative = one tools knob forced" % ("forward" if fwd else "backward"))
    print("# (set_own 1/2: owner windows wherever possible / never; set_ww_big 1/2: one fat row-group workgroup per CU always / never;")
    print("#  force_ring 1/2: register loops / LDS-DMA ring; set_seg_min_div 1: segment walk for whole spans only;")
    print("#  forward only: set_fwd_direct 1/2/4: lanes read their own parameters on the usual grid / the LDS table / direct for every lane form)")
    worst = []
    for dt_name in ("f32", "bf16"):
        dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt_name]
        for shape, axis in shapes(rng, count):
            base = time_bwd(shape, dtype, (("policy", 0),), "set_own", axis=axis, fwd=fwd)["policy"]
            best = None
            for knob, v in (ALTS_FWD if fwd else ALTS):
                r = time_bwd(shape, dtype, (("alt", v),), knob, axis=axis, fwd=fwd)["alt"]
                if (fwd or r[1] != base[1]) and (best is None or r[0] < best[0]):      # (the forward's note does not tell table from direct)
                    best = (r[0], "%s %d" % (knob, v), r[1])
            line = "%-4s %-22s axis %d  policy %7.1f [%s]" % (dt_name, "x".join(map(str, shape)), axis, base[0], base[1])
            if best is not None:
                gain = best[0] / base[0] - 1
                line += "   best other %7.1f [%s: %s] %+5.1f %%%s" % (best[0], best[1], best[2], 100 * gain, "   <-- policy is behind" if gain < -0.07 else "")
                if gain < -0.07:
                    worst.append(line)
            print(line, flush=True)
    print("# %d shape(s) where a forced alternative is more than 7 %% ahead of the policy" % len(worst))


if __name__ == "__main__":
    main()
