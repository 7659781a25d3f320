#!/usr/bin/env python3
"""How short may an owner's run be?  (lsq_pc_geom.hpp plan_own, kOwnMinRunBytes)  NCHW-style activations whose channel rows
are a few positions long: the 256-lane windows + finalize against owner windows planned with a minimum run of 1 (the smallest
packet-aligned channel group, the first form of the plan), 256, 512 and 1024 bytes; backward op, cold inputs, HIP-graph replay.
Output: profiles/r04_owner_min_run.txt."""
import sys

import torch

from exp_knob_ab import time_bwd
import lsq_tools

SHAPES = [(512, 2048, 7), (256, 2048, 7), (64, 1024, 3, 3), (128, 2048, 3, 3), (64, 2048, 4, 4), (128, 2048, 4, 4), (64, 1024, 5, 5),
          (128, 1024, 5, 5), (64, 1024, 6, 6), (64, 2048, 6, 6), (32, 1024, 8, 8), (64, 1024, 8, 8), (64, 512, 10, 10), (64, 2048, 7, 7)]


def main():
    print("# tools/exp_owner_min_run.py: backward op, us, cold inputs; win = 256-lane windows + finalize (set_own 2); own >= B = owner windows")
    print("# forced wherever the plan allows (set_own 1) with runs of at least B bytes (set_own_min_run); the launch each setting produced in brackets")
    for dt_name in sys.argv[1:] or ["f32", "bf16"]:
        dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt_name]
        for shape in SHAPES:
            cells = []
            for label, own, mr in (("win", 2, 0), ("own >= 1", 1, 1), ("own >= 256", 1, 256), ("own >= 512", 1, 512), ("own >= 1024", 1, 1024)):
                lsq_tools.set_knob("set_own_min_run", mr)
                r = time_bwd(shape, dtype, ((label, own),), "set_own", axis=1)
                t, note = r[label]
                cells.append("%s %6.1f [%s]" % (label, t, note.split(" lanes")[0]))
            lsq_tools.set_knob("set_own_min_run", 0)
            print("%-4s %-16s %s" % (dt_name, "x".join(str(d) for d in shape), "  |  ".join(cells)), flush=True)


if __name__ == "__main__":
    main()
