#!/usr/bin/env python3
"""Row-group windows (quantized axis last, [rows, C]) at 8-67 M elements: register loops, the ring in the usual 256-lane
workgroups, and the ring in ONE 768/1024-lane workgroup per CU; backward op, cold inputs, HIP-graph replay.
Output: profiles/r04_rowgroup_mid.txt."""
import torch

from exp_knob_ab import time_bwd
import lsq_tools

print("# tools/exp_rowgroup_mid.py: backward op, us; pol = the policy; reg = register loops (force_ring 1); ring = LDS-DMA ring in 256-lane workgroups")
print("# (force_ring 2 + set_ww_big 2); big = the ring in one fat workgroup per CU (force_ring 2 + set_ww_big 1)")
for dt_name in ("f32", "bf16"):
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt_name]
    for C in (64, 128, 256, 512, 768, 1024, 1536, 2048, 4096):
        for target in (3 << 22, 1 << 24, 1 << 25, 1 << 26):
            rows = target // C
            shape = (rows, C)
            res = {}
            res["pol"] = time_bwd(shape, dtype, (("x", 0),), "force_ring", axis=1)["x"]
            res["reg"] = time_bwd(shape, dtype, (("x", 1),), "force_ring", axis=1)["x"]
            lsq_tools.set_knob("set_ww_big", 2)
            res["ring"] = time_bwd(shape, dtype, (("x", 2),), "force_ring", axis=1)["x"]
            lsq_tools.set_knob("set_ww_big", 1)
            res["big"] = time_bwd(shape, dtype, (("x", 2),), "force_ring", axis=1)["x"]
            lsq_tools.set_knob("set_ww_big", 0)
            best = min(("reg", "ring", "big"), key=lambda k: res[k][0])
            print("%-4s %-14s %9d el  pol %6.1f [%s]  reg %6.1f  ring %6.1f [%s]  big %6.1f [%s]   best %-4s %+5.1f %% vs policy" % (
                dt_name, "%dx%d" % shape, rows * C, res["pol"][0], res["pol"][1], res["reg"][0], res["ring"][0], res["ring"][1].split(" lanes")[0],
                res["big"][0], res["big"][1].split(" lanes")[0], best, (res[best][0] / res["pol"][0] - 1) * 100), flush=True)
