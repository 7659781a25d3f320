import sys, os, time
sys.path.insert(0, "lsqfakequantize-pytorch_amd")
import torch, torchlsq
from torchlsq import synth
from torchlsq.functional import lsq
dev = torch.device("cuda:0")
x, g, scale, shift = synth.make_inputs("cfg1", device=dev, dtype=torch.float32)
xs = x.clone().requires_grad_(True); ss = scale.clone().requires_grad_(True); bs = shift.clone().requires_grad_(True)
N = 2000
from torchlsq import extension
def native():
    extension.set_host_binding("native")
    for _ in range(N):
        y = lsq(xs, ss, bs, 0, 127, 0, 255); y.backward(g)
def direct():
    extension.set_host_binding("ctypes")
    for _ in range(N):
        y = lsq(xs, ss, bs, 0, 127, 0, 255); y.backward(g)
wq = torch.randn(64, 64, 3, 3, device=dev, requires_grad=True)
wsc = torch.full((64,), 0.01, device=dev, requires_grad=True); wsh = torch.zeros(64, device=dev, requires_grad=True)
gw = torch.randn_like(wq)
def native_pc():
    extension.set_host_binding("native")
    for _ in range(N):
        y = lsq(wq, wsc, wsh, -128, 127, -128, 127, 0, True, 1.0, False, True); y.backward(gw)
def direct_pc():
    extension.set_host_binding("ctypes")
    for _ in range(N):
        y = lsq(wq, wsc, wsh, -128, 127, -128, 127, 0, True, 1.0, False, True); y.backward(gw)
def disp():
    for _ in range(N):
        y = torch.ops.torchlsq.lsq(xs, ss, bs, 0, 127, 0, 255, 1, True, 1.0, True, False, False, False); y.backward(g)
def evalm():
    for _ in range(N):
        y = lsq(xs, ss, bs, 0, 127, 0, 255, eval_mode=True); y.backward(g)
def plain():
    for _ in range(N):
        y = xs * 2.0; y.backward(g)
best = {}
for rep in range(6):
    for name, fn in (("native (C++ binding)", native), ("direct (ctypes)", direct), ("native per-channel weight", native_pc),
                     ("ctypes per-channel weight", direct_pc), ("dispatcher", disp), ("eval(masked)", evalm), ("plain mul autograd", plain)):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / N * 1e6
        best[name] = min(best.get(name, 1e9), us)
for name, us in best.items():
    print("%-28s best of 6: %.1f us per forward+backward" % (name, us))
