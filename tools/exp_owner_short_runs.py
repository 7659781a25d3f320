#!/usr/bin/env python3
"""Owner windows on channel rows of a few positions (short runs), 3.5-17 M elements: the plan's fattest channel group (one
owner per CU, forced: set_own 1) against the 256-lane windows + finalize; backward op, cold inputs, HIP-graph replay.
Output: profiles/r04_owner_short_runs.txt."""
import torch

from exp_knob_ab import time_bwd

SHAPES = [(384, 2048, 7), (512, 2048, 7), (768, 2048, 7), (320, 2048, 8), (512, 2048, 8), (768, 2048, 8), (1024, 2048, 8), (256, 2048, 4, 4), (384, 2048, 4, 4),
          (128, 2048, 5, 5), (256, 1024, 5, 5), (192, 1024, 8, 8), (128, 2048, 6, 6), (96, 1024, 10, 10), (128, 1024, 10, 10), (192, 2048, 3, 3)]

print("# tools/exp_owner_short_runs.py: backward op, us, cold inputs; win = 256-lane windows + finalize (set_own 2), own = owner windows forced (set_own 1)")
for dt_name in ["f32", "bf16"]:
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt_name]
    for shape in SHAPES:
        r = time_bwd(shape, dtype, (("win", 2), ("own", 1)), "set_own", axis=1)
        n = 1
        for d in shape:
            n *= d
        print("%-4s %-16s %9d el  win %6.1f [%s]  own %6.1f [%s]  own/win %+5.1f %%" % (
            dt_name, "x".join(map(str, shape)), n, r["win"][0], r["win"][1].split(" lanes")[0], r["own"][0], r["own"][1].split(" lanes")[0],
            (r["own"][0] / r["win"][0] - 1) * 100), flush=True)
