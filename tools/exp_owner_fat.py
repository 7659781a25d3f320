#!/usr/bin/env python3
"""Which channel group should an owner take?  (lsq_pc_geom.hpp plan_own)  The smallest packet-aligned group (most owners, the
first form of the plan) against the LARGEST group that still gives every CU an owner (fewer, fatter owners with longer runs),
next to the 256-lane windows + finalize; backward op, cold inputs, HIP-graph replay.  Output: profiles/r04_owner_fat_ab.txt."""
import sys

import torch

from exp_knob_ab import time_bwd
import lsq_tools

SHAPES = [(64, 2048, 7, 7), (32, 2048, 7, 7), (16, 2048, 7, 7), (80, 2048, 7, 7), (32, 1024, 14, 14), (16, 1024, 14, 14), (16, 512, 28, 28),
          (32, 256, 28, 28), (64, 256, 14, 14), (128, 256, 14, 14), (64, 1024, 7, 7), (128, 1024, 7, 7), (32, 512, 14, 14), (64, 512, 14, 14),
          (64, 2048, 4, 4), (128, 2048, 4, 4), (64, 1024, 5, 5), (128, 1024, 5, 5), (32, 1024, 8, 8), (64, 1024, 8, 8), (64, 512, 10, 10),
          (256, 2048, 7), (256, 2048, 7, 7), (128, 1024, 14, 14)]


def main():
    print("# tools/exp_owner_fat.py: backward op, us, cold inputs; win = 256-lane windows + finalize (set_own 2); own = owner windows forced")
    print("# (set_own 1) with the smallest channel group (set_own_fat 1) / the largest that still gives every CU an owner (set_own_fat 2)")
    for dt_name in sys.argv[1:] or ["f32", "bf16"]:
        dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt_name]
        for shape in SHAPES:
            cells, ts = [], {}
            for label, own, fat in (("win", 2, 0), ("own, smallest", 1, 1), ("own, fattest", 1, 2)):
                lsq_tools.set_knob("set_own_fat", fat)
                r = time_bwd(shape, dtype, ((label, own),), "set_own", axis=1)
                t, note = r[label]
                ts[label] = t
                cells.append("%s %6.1f [%s]" % (label, t, note.split(" lanes")[0]))
            lsq_tools.set_knob("set_own_fat", 0)
            print("%-4s %-16s %s   fattest / smallest %+5.1f %%" % (dt_name, "x".join(str(d) for d in shape), "  |  ".join(cells),
                                                                  (ts["own, fattest"] / ts["own, smallest"] - 1) * 100), flush=True)


if __name__ == "__main__":
    main()
