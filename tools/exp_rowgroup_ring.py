#!/usr/bin/env python3
"""Row-group windows (quantized axis last, [rows, C]): LDS-DMA ring against register loops on SMALL tensors, where the policy's
tiles-per-workgroup floor decides; backward op, cold inputs, HIP-graph replay.  Output: profiles/r04_rowgroup_ring_small.txt."""
import torch

from exp_knob_ab import time_bwd

print("# tools/exp_rowgroup_ring.py: backward op, us; pol = the policy, reg = register loops (force_ring 1), ring = LDS-DMA ring (force_ring 2)")
for dt_name in ("f32", "bf16"):
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt_name]
    for C in (64, 128, 256, 384, 512, 768, 1024, 2048, 4096):
        for target in (1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23):
            rows = max(8, target // C)
            shape = (rows, C)
            r = time_bwd(shape, dtype, (("pol", 0), ("reg", 1), ("ring", 2)), "force_ring", axis=1)
            best = min(("reg", "ring"), key=lambda k: r[k][0])
            print("%-4s %-14s %9d el  pol %6.1f [%s]  reg %6.1f [%s]  ring %6.1f [%s]   best %s %+5.1f %% vs policy" % (
                dt_name, "%dx%d" % shape, rows * C, r["pol"][0], r["pol"][1], r["reg"][0], r["reg"][1].split(" of")[0], r["ring"][0],
                r["ring"][1].split(" of")[0], best, (r[best][0] / r["pol"][0] - 1) * 100), flush=True)
