#!/usr/bin/env python3
"""How far d_scale / d_shift are from the reference, per BASELINE config, storage type and device path.

For every digest of tests/golden/config_digests.json (expected values = the reference CPU csrc's own outputs): run the op
on the GPU (and the small configs on the CPU kernels too) and print
    max |got - ref| / |ref|            the plain relative error (meaningful where the sum has no cancellation)
    max |got - ref| / sum|terms|       what tests/helpers.py::assert_reduction_close bounds by 1e-6
next to the same two figures for the reference itself against the exact (fp64) sum of its own fp32 terms -- the reference
sums in fp32 with at::sum, so it sits up to ~9e-7 relative from the exact sum even without cancellation, and orders of
magnitude further where the terms cancel.  Output: profiles/r03_reduction_margin.txt."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torchlsq  # noqa: E402,F401
from torchlsq import synth  # noqa: E402
from torchlsq.functional import lsq  # noqa: E402


def figures(got, ref, abs_terms):
    got, ref, ab = (np.asarray(a, dtype=np.float64).reshape(-1) for a in (got, ref, abs_terms))
    nz = np.abs(ref) > 0
    if not nz.any():
        return None
    err = np.abs(got - ref)[nz]
    return (err / np.abs(ref[nz])).max(), (err / np.maximum(ab[nz], 1e-300)).max(), (ab[nz] / np.abs(ref[nz])).max()


def run(d, device, dtype):
    x, g, scale, shift = synth.make_inputs(d["config"], device=device, dtype=dtype, abs_grad=d["abs_grad"])
    x.requires_grad_(True); scale.requires_grad_(True); shift.requires_grad_(True)
    lsq(x, scale, shift, **synth.op_kwargs(d["config"])).backward(g)
    ds = scale.grad.cpu().numpy()
    db = shift.grad.cpu().numpy() if shift.grad is not None else np.zeros_like(ds)
    return ds, db


def main():
    digests = json.load(open(os.path.join(ROOT, "tests", "golden", "config_digests.json")))["configs"]
    dev = torch.device("cuda:0")
    print("# tools/exp_reduction_margin.py: d_scale / d_shift against the reference CPU csrc's outputs (tests/golden/config_digests.json)")
    print("# columns: max|got-ref|/|ref|   max|got-ref|/sum|terms|   (cancellation: max sum|terms|/|sum|)   -- 'reference vs exact' = the")
    print("# reference's own fp32 at::sum against the fp64 sum of its fp32 terms (oracle), the floor any comparison with it inherits")
    print("%-18s %-5s %-9s %-3s %12s %12s %10s" % ("config", "store", "path", "", "rel", "/sum|terms|", "cancel"))
    for name, d in digests.items():
        big = int(np.prod(d["shape"])) > (1 << 26)
        storage = torch.bfloat16 if d.get("bf16_io") else torch.float32
        rows = [("gpu", dev)]
        if not big and storage == torch.float32:
            rows.append(("cpu-lib", torch.device("cpu")))
        for q, wide, ab in (("ds", "oracle_ds_wide", "oracle_abs_ds"), ("db", "oracle_db_wide", "oracle_abs_db")):
            f = figures(d[wide], d[q], d[ab])
            if f:
                print("%-18s %-5s %-9s %-3s %12.3g %12.3g %10.3g   (reference vs exact)" % (name, "f32", "reference", q, f[0], f[1], f[2]))
        for label, device in rows:
            ds, db = run(d, device, storage)
            for q, got, ab in (("ds", ds, "oracle_abs_ds"), ("db", db, "oracle_abs_db")):
                f = figures(got, d[q], d[ab])
                if f:
                    print("%-18s %-5s %-9s %-3s %12.3g %12.3g %10.3g" % (name, "bf16" if storage == torch.bfloat16 else "f32", label, q, f[0], f[1], f[2]))
                f = figures(got, d["oracle_%s_wide" % q], d[ab])
                if f:
                    print("%-18s %-5s %-9s %-3s %12.3g %12.3g %10s   (against the exact sum)" % ("", "", label, q, f[0], f[1], ""))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
