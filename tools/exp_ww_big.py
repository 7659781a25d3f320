#!/usr/bin/env python3
"""Row-group windows of mid-sized last-axis tensors: the usual 3-4-wave workgroups (knob 2) against one 768/1024-lane
workgroup per CU (knob 1) and the shipped policy (knob 0); GPU-side us per backward incl. finalize, (windows x splits) in
brackets, results checked against each other (dx bits, d_scale/d_shift to 1e-6 of the sum of |terms| scale)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torchlsq  # noqa: F401
from torchlsq import extension as E, synth
import lsq_tools  # noqa: E402  (tools build of the library: `_ex` entry points, lsq_hip_debug_* knobs)
lsq_tools.activate()
lib = E.library()
lib.lsq_hip_debug_set_ww_big.argtypes = [ctypes.c_int]
lib.lsq_hip_debug_last_launch.argtypes = [ctypes.POINTER(ctypes.c_int * 8)]
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps):
                fn()
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


SHAPES = (((64, 197, 768), 2), ((16, 197, 768), 2), ((256, 197, 768), 2), ((32, 197, 1024), 2), ((8192, 4096), 1), ((4096, 1024), 1),
          ((2048, 4096), 1), ((16, 56, 56, 256), 3), ((64, 56, 56, 256), 3), ((32, 1024, 384), 2), ((8, 512, 1280), 2))
for shape, axis in SHAPES:
    for dt in (torch.float32, torch.bfloat16):
        n = 1
        for d in shape: n *= d
        x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt).view(shape)
        g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt).view(shape)
        C = shape[axis]
        s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev); b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        q = (0, 127, 0, 255, True, 1.0, False, False, False)
        res, outs = [], {}
        for knob in (2, 1, 0, 2, 1):
            lib.lsq_hip_debug_set_ww_big(knob)
            E._WS_BYTES_PC.clear()
            outs[knob] = E.hip_backward_per_channel(g, x, s, b, axis, *q)
            t = timeit(lambda: E.hip_backward_per_channel(g, x, s, b, axis, *q))
            o = (ctypes.c_int * 8)(); lib.lsq_hip_debug_last_launch(ctypes.byref(o))
            res.append("%s %.1f us (%dx%d)" % ({2: "usual", 1: "big", 0: "policy"}[knob], t, o[0], o[1]))
        lib.lsq_hip_debug_set_ww_big(0)
        E._WS_BYTES_PC.clear()
        a, c = outs[2], outs[1]
        same_dx = torch.equal(a[0], c[0])
        scale = a[1].abs().max().item() + 1e-30
        err = max((a[1] - c[1]).abs().max().item(), (a[2] - c[2]).abs().max().item()) / scale
        print("%-9s %-18s %s | dx equal %s, ds/db rel diff %.1e" % (str(dt).replace("torch.", ""), shape, " | ".join(res), same_dx, err),
              flush=True)
