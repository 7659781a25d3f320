#!/usr/bin/env python3
"""Time the CPU baseline on this box's host cores (bench.py's `cpu_baseline` leg; TEST/BENCH TOOLING).

Preferred: the REFERENCE's own CPU csrc -- oracle/_ref/libtorchlsq_ref_ops.so, built by
oracle/build_ref.py from /root/reference in the build container and shipped with the snapshot
(kind "reference").  Fallback: the C restatement oracle/lsq_oracle.c with OpenMP (kind "port").
Runs in its own process (the reference library registers the same `torchlsq::*` op names as the
product) and never touches the GPU.  Prints one JSON line.

The sample is the workload's full shape, timed at all host cores and at ONE thread, each
bounded: passes are repeated until `--budget` seconds of CPU wall time are spent (at least one pass).  bf16 workloads are
timed on the fp32 stand-in (the reference's CPU path has no bf16: AT_DISPATCH_FLOATING_TYPES, lsq_cpu.cpp:37,92).
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

# bench.py workload -> (synth config, channel axis override)
WORKLOADS = {"cfg1": ("cfg1", None), "cfg2": ("cfg2", None), "cfg3": ("cfg3", None), "cfg4": ("cfg4", None),
             "cfg5": ("cfg5", None), "cfg5_bf16": ("cfg5", None), "cfg5_axis0": ("cfg5", 0),
             "tok": ("tok", None), "tok_bf16": ("tok", None), "vit": ("vit", None), "vit_bf16": ("vit", None)}


def load_synth():
    p = os.path.join(ROOT, "lsqfakequantize-pytorch_amd", "torchlsq", "synth.py")
    spec = importlib.util.spec_from_file_location("_synth_by_path", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def timed(step, budget_s, max_reps):
    times = []
    while len(times) < max_reps and (not times or sum(times) + min(times) <= budget_s):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    return times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default=None, help="default: the workload's configured shape")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--budget", type=float, default=12.0, help="seconds of timed CPU wall per thread setting")
    ap.add_argument("--kind", default="auto", choices=["auto", "reference", "port"])
    a = ap.parse_args()
    import torch
    S = load_synth()
    cfg, axis_override = WORKLOADS[a.workload]
    c = dict(S.CONFIGS[cfg])
    if axis_override is not None:
        c["axis"] = axis_override
    shape = tuple(int(v) for v in a.shape.split(",")) if a.shape else tuple(c["shape"])
    x, g, scale, shift = S.make_inputs(c, dtype=torch.float32, shape=shape)
    n = x.numel()
    q = (c["qmin"], c["qmax"], c["tmin"], c["tmax"])
    sym = not c["affine"]
    pc, axis = c["per_channel"], c["axis"]
    tail = q + (True, 1.0, sym, False, False)
    ref_so = os.path.join(HERE, "_ref", "libtorchlsq_ref_ops.so")
    kind = a.kind
    if kind == "auto":
        kind = "reference" if os.path.isfile(ref_so) else "port"
    err = None
    all_cores = torch.get_num_threads()
    if kind == "reference":
        try:
            torch.ops.load_library(ref_so)
            ops = torch.ops.torchlsq

            def step():
                if pc:
                    ops.lsq_forward_per_channel(x, scale, shift, axis, *tail)
                    ops.lsq_backward_per_channel(g, x, scale, shift, axis, *tail)
                else:
                    ops.lsq_forward_per_tensor(x, scale, shift, *tail)
                    ops.lsq_backward_per_tensor(g, x, scale, shift, *tail)

            def set_threads(k):
                torch.set_num_threads(k)
            step()
        except Exception as e:       # e.g. a different libtorch on the box
            err = repr(e)[:300]
            kind = "port"
    if kind == "port":
        from oracle import lsq_oracle as O
        all_cores = O.max_threads()
        xn, gn = x.numpy(), g.numpy()
        sn, bn = scale.numpy(), shift.numpy()
        s0, b0 = scale[0].item(), shift[0].item()
        outer, C, inner = O.axis_to_ocl(shape, axis) if pc else (1, 1, n)

        def step():
            if pc:
                O.fwd_pc(xn, sn, bn, outer, C, inner, *q)
                O.bwd_pc(gn, xn, sn, bn, outer, C, inner, *q, True, 1.0, sym, want_buffers=True)
            else:
                O.fwd_pt(xn, s0, b0, *q)
                O.bwd_pt(gn, xn, s0, b0, *q, True, 1.0, sym, want_buffers=True)   # materialise ds/db buffers like lsq_cpu.cpp:80-82

        def set_threads(k):
            O.set_threads(k)
        step()
    set_threads(all_cores)
    t_all = timed(step, a.budget, 10)
    set_threads(1)
    t_one = timed(step, a.budget, 3)
    best = min(t_all)
    out = {"value": round(n / best / 1e9, 4), "unit": "GElem/s", "cores": int(all_cores), "kind": kind,
           "sample": "%s fp32 %s (%d elements, the workload's full shape%s): best of %d fwd+bwd passes at %d threads (%.1f s), "
                     "and best of %d at 1 thread (%.1f s)"
                     % (cfg, list(shape), n, "; fp32 stand-in for the bf16 storage" if a.workload.endswith("bf16") else "",
                        len(t_all), all_cores, sum(t_all), len(t_one), sum(t_one)),
           "ms_per_step": round(best * 1e3, 2),
           "one_thread": {"value": round(n / min(t_one) / 1e9, 4), "unit": "GElem/s", "cores": 1,
                          "ms_per_step": round(min(t_one) * 1e3, 2)}}
    if err:
        out["reference_load_error"] = err
    print(json.dumps(out))


if __name__ == "__main__":
    main()
