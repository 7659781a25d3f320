#!/usr/bin/env python3
"""Host time of one LSQFakeQuantizer call (forward + backward, steady state) against functional.lsq (diagnostic)."""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch
import torchlsq
from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
from torchlsq.functional import lsq
from torchlsq.quantized import LSQFakeQuantizer
dev = torch.device("cuda:0")
act = LSQFakeQuantizer(MovingAverageMinMaxObserver, "activation", init_batches=1).to(dev).train()
wq = LSQFakeQuantizer(MovingAveragePerChannelMinMaxObserver, "weight", dtype=torch.qint8, qscheme=torch.per_channel_symmetric).to(dev).train()
x = torch.rand(4, 64, 56, 56, device=dev, requires_grad=True)
w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
gx, gw = torch.randn_like(x), torch.randn_like(w)
for _ in range(4):
    (act(x).sum() + wq(w).sum()).backward()
s, b = act.scale, act.shift
N = 2000
def mod_act():
    for _ in range(N): act(x).backward(gx)
def mod_w():
    for _ in range(N): wq(w).backward(gw)
def fn_act():
    for _ in range(N): lsq(x, s, b, 0, 127, 0, 255).backward(gx)
def plain():
    for _ in range(N): (x * 2.0).backward(gx)
best = {}
for rep in range(5):
    for name, fn in (("module activation", mod_act), ("module weight (per-channel)", mod_w), ("functional.lsq", fn_act), ("plain mul", plain)):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best[name] = min(best.get(name, 1e9), (time.perf_counter() - t0) / N * 1e6)
for k, v in best.items():
    print("%-30s %.1f us per forward+backward" % (k, v))
pr = cProfile.Profile(); pr.enable(); mod_act(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
