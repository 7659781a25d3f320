#!/usr/bin/env python3
"""Print the grid the window-mode backward chose for a few shapes (diagnostic): windows x splits, the kernel's register
count as the runtime reports it and the residency (workgroups per CU) the geometry was sized for."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torchlsq  # noqa: F401
from torchlsq import extension as E, synth
import lsq_tools  # noqa: E402  (tools build of the library: `_ex` entry points, lsq_hip_debug_* knobs)
lsq_tools.activate()

lib = E.library()
lib.lsq_hip_debug_last_launch.argtypes = [ctypes.POINTER(ctypes.c_int * 8)]
dev = torch.device("cuda:0")
for dt in (torch.float32, torch.bfloat16):
    for shape, axis in (((256, 2048, 7, 7), 1), ((8192, 4096), 1), ((64, 197, 768), 2), ((64, 56, 56, 256), 3), ((32, 256, 56, 56), 1)):
        n = 1
        for d in shape:
            n *= d
        x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt).view(shape)
        g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt).view(shape)
        C = shape[axis]
        s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev)
        b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        E.hip_backward_per_channel(g, x, s, b, axis, 0, 127, 0, 255, True, 1.0, False, False, False)
        torch.cuda.synchronize()
        out = (ctypes.c_int * 8)()
        lib.lsq_hip_debug_last_launch(ctypes.byref(out))
        print("%-9s %-20s axis %d: grid %d x %d = %d workgroups, numRegs %d, sized for %d resident per CU" %
              (str(dt).replace("torch.", ""), shape, axis, out[0], out[1], out[0] * out[1], out[3], out[2]))
