#!/usr/bin/env python3
"""Time the CPU baseline on this box's host cores (bench.py's `cpu_baseline` leg; TEST/BENCH TOOLING).

Preferred: the REFERENCE's own CPU csrc -- oracle/_ref/libtorchlsq_ref_ops.so, built by
oracle/build_ref.py from /root/reference in the build container and shipped with the snapshot
(kind "reference").  Fallback: the C restatement oracle/lsq_oracle.c with OpenMP (kind "port").
Runs in its own process (the reference library registers the same `torchlsq::*` op names as the
product) and never touches the GPU.  Prints one JSON line.
"""
import argparse
import importlib.util
import json
import os
import sys
import time
import warnings

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def load_synth():
    p = os.path.join(ROOT, "lsqfakequantize-pytorch_amd", "torchlsq", "synth.py")
    spec = importlib.util.spec_from_file_location("_synth_by_path", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="16,512,56,56")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--kind", default="auto", choices=["auto", "reference", "port"])
    a = ap.parse_args()
    import torch
    S = load_synth()
    shape = tuple(int(v) for v in a.shape.split(","))
    c = S.CONFIGS[a.config]
    x, g, scale, shift = S.make_inputs(a.config, dtype=torch.float32, shape=shape)
    n = x.numel()
    q = (c["qmin"], c["qmax"], c["tmin"], c["tmax"])
    sym = not c["affine"]
    ref_so = os.path.join(HERE, "_ref", "libtorchlsq_ref_ops.so")
    kind = a.kind
    if kind == "auto":
        kind = "reference" if os.path.isfile(ref_so) else "port"
    err = None
    if kind == "reference":
        try:
            torch.ops.load_library(ref_so)
            ops = torch.ops.torchlsq
            cores = torch.get_num_threads()

            def step():
                ops.lsq_forward_per_tensor(x, scale, shift, *q, True, 1.0, sym, False, False)
                ops.lsq_backward_per_tensor(g, x, scale, shift, *q, True, 1.0, sym, False, False)
            step()
        except Exception as e:       # e.g. a different libtorch on the box
            err = repr(e)[:300]
            kind = "port"
    if kind == "port":
        from oracle import lsq_oracle as O
        cores = O.max_threads()
        xn, gn = x.numpy(), g.numpy()
        s0, b0 = scale[0].item(), shift[0].item()

        def step():
            O.fwd_pt(xn, s0, b0, *q)
            O.bwd_pt(gn, xn, s0, b0, *q, True, 1.0, sym, want_buffers=True)   # materialise ds/db buffers like lsq_cpu.cpp:80-82
        step()
    times = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    best = min(times)
    out = {"value": round(n / best / 1e9, 4), "unit": "GElem/s", "cores": int(cores), "kind": kind,
           "sample": "%s fp32 %s (%d elements), best of %d fwd+bwd passes, %.2f s CPU wall total"
                     % (a.config, list(shape), n, a.reps, sum(times)),
           "ms_per_step": round(best * 1e3, 2)}
    if err:
        out["reference_load_error"] = err
    print(json.dumps(out))


if __name__ == "__main__":
    main()
