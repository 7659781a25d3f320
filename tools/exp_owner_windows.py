#!/usr/bin/env python3
"""Round-4 experiment (c): OWNER windows of the per-channel backward (lsq_pc_geom.hpp plan_own: a fat workgroup owns k whole
channels for all rows; no partials, no finalize launch) against the 256-lane windows + finalize launch, backward op, cold
inputs, HIP-graph replay; owner windows with and without the waves' turns at the higher issue priority.  Output: profiles/r04_owner_windows_ab.txt."""
import sys

import torch

from exp_knob_ab import lib, time_bwd
import lsq_tools

SHAPES = [(256, 2048, 7, 7), (128, 1024, 14, 14), (128, 512, 28, 28), (64, 2048, 14, 14), (64, 512, 28, 28),            # 25-51 M elements
          (192, 2048, 7, 7), (96, 1024, 14, 14), (48, 512, 28, 28), (160, 2048, 7, 7), (80, 1024, 14, 14), (40, 512, 28, 28),  # 16-19 M
          (128, 2048, 7, 7), (64, 1024, 14, 14), (32, 2048, 14, 14), (32, 512, 28, 28),                                  # 12.8 M
          (96, 2048, 7, 7), (80, 2048, 7, 7), (48, 1024, 14, 14),                                                        # 8-10 M
          (64, 2048, 7, 7), (32, 2048, 7, 7), (16, 2048, 7, 7), (32, 1024, 14, 14), (16, 1024, 14, 14),                  # <= 2^23: the policy's band
          (16, 512, 28, 28), (32, 256, 28, 28), (64, 256, 14, 14), (128, 256, 14, 14),
          (58, 2048, 7, 7), (83, 2048, 7, 7), (33, 2048, 7, 7), (83, 512, 14, 14), (37, 1024, 14, 14),                   # row counts without a divisor near the row slots: a short last tile
          (64, 512, 7, 7), (32, 256, 56, 56)]                                                                            # not eligible: windows either way


def main():
    print("# tools/exp_owner_windows.py: backward op (kernel + finalize launch where there is one), us, cold inputs; own: 2 = 256-lane windows,")
    print("# 1 = owner windows (at most 512 lanes), 3 = owner windows without the waves' turns at the higher issue priority;")
    print("# the ring copies' streaming hint by the policy (on above 32 MB)")
    for dt_name in sys.argv[1:] or ["bf16", "f32"]:
        dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt_name]
        for shape in SHAPES:
            r = time_bwd(shape, dtype, (("win", 2), ("own", 1), ("own, no turns", 3)), "set_own", axis=1)
            row = "  ".join("%s %6.1f (%s)" % (k, r[k][0], r[k][1].split(" of ")[0] + " of " + r[k][1].split(" of ")[1].split(",")[0])
                            for k in ("win", "own", "own, no turns"))
            print("%-4s %-20s %s   own / win %+5.1f %%" % (dt_name, "x".join(str(d) for d in shape), row, (r["own"][0] / r["win"][0] - 1) * 100), flush=True)


if __name__ == "__main__":
    main()
