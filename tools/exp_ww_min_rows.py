#!/usr/bin/env python3
"""Row-group-window backward (last-axis shapes): how few rows may a workgroup walk?  Sweep of the minimum rows per
workgroup (its partial row costs 16 bytes per slot) x workgroups per CU; GPU-side us per backward incl. finalize."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.argv = sys.argv[:1]
import torch
import torchlsq  # noqa: F401
from torchlsq import extension as E, synth
import lsq_tools  # noqa: E402  (tools build of the library: `_ex` entry points, lsq_hip_debug_* knobs)
lsq_tools.activate()
lib = E.library()
lib.lsq_hip_debug_set_ww_min_rows.argtypes = [ctypes.c_int]
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


for shape, axis in (((64, 197, 768), 2), ((8192, 4096), 1), ((64, 56, 56, 256), 3), ((65536, 1024), 1)):
    for dt in (torch.float32, torch.bfloat16):
        n = 1
        for d in shape: n *= d
        x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt).view(shape)
        g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt).view(shape)
        C = shape[axis]
        s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev); b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        q = (0, 127, 0, 255, True, 1.0, False, False, False)
        out = []
        for mr in (0, 16, 8, 4):
            lib.lsq_hip_debug_set_ww_min_rows(mr)
            E._WS_BYTES_PC.clear()
            row = []
            for bpc in (4, 8, 16):
                for dma in (1, 2):
                    base = ((4 | (3 << 8)) if dt == torch.float32 else (1 | (3 << 8) | (1 << 10))) | (bpc << 16) | (dma << 12)
                    t = timeit(lambda: E.hip_backward_per_channel(g, x, s, b, axis, *q, variant=base))
                    o = (ctypes.c_int * 8)(); lib.lsq_hip_debug_last_launch(ctypes.byref(o))
                    row.append("%d/CU %s %.1f(%dx%d)" % (bpc, "reg" if dma == 1 else "dma", t, o[0], o[1]))
            out.append("min_rows %s: %s" % (mr or "rule", "  ".join(row)))
        lib.lsq_hip_debug_set_ww_min_rows(0)
        E._WS_BYTES_PC.clear()
        print("%-9s %-18s\n   %s" % (str(dt).replace("torch.", ""), shape, "\n   ".join(out)), flush=True)
