#!/usr/bin/env python3
"""Run forward + backward of ONE per-channel shape a few times (to be wrapped in rocprofv3 --kernel-trace --stats).
usage: python3 tools/exp_one_shape.py 64,197,768 2 float32 [ww_big knob: 0 policy / 1 always / 2 never]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torchlsq  # noqa: F401
from torchlsq import synth
shape = tuple(int(v) for v in sys.argv[1].split(","))
axis = int(sys.argv[2])
dt = getattr(torch, sys.argv[3]) if len(sys.argv) > 3 else torch.float32
dev = torch.device("cuda:0")
if len(sys.argv) > 4:
    import ctypes
    from torchlsq import extension as E
    import lsq_tools  # noqa: E402  (tools build of the library: `_ex` entry points, lsq_hip_debug_* knobs)
    lsq_tools.activate()
    E.library().lsq_hip_debug_set_ww_big.argtypes = [ctypes.c_int]
    E.library().lsq_hip_debug_set_ww_big(int(sys.argv[4]))
n = 1
for d in shape: n *= d
x = synth.normal_like(n, 1, 0.5, 1.0, device=dev, dtype=dt).view(shape)
g = synth.normal_like(n, 2, 0.0, 1e-3, device=dev, dtype=dt).view(shape)
C = shape[axis]
s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev); b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
ops = torch.ops.torchlsq
for _ in range(30):
    ops.lsq_forward_per_channel(x, s, b, axis, 0, 127, 0, 255, True, 1.0, False, False, False)
    ops.lsq_backward_per_channel(g, x, s, b, axis, 0, 127, 0, 255, True, 1.0, False, False, False)
torch.cuda.synchronize()
