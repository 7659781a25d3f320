import sys, os, time
sys.path.insert(0, "lsqfakequantize-pytorch_amd")
import torch, torchlsq
from torchlsq import synth
from torchlsq.functional import lsq
dev = torch.device("cuda:0")
x, g, scale, shift = synth.make_inputs("cfg1", device=dev, dtype=torch.float32)
xs = x.clone().requires_grad_(True); ss = scale.clone().requires_grad_(True); bs = shift.clone().requires_grad_(True)
N = 2000
def direct():
    for _ in range(N):
        y = lsq(xs, ss, bs, 0, 127, 0, 255); y.backward(g)
def disp():
    for _ in range(N):
        y = torch.ops.torchlsq.lsq(xs, ss, bs, 0, 127, 0, 255, 1, True, 1.0, True, False, False, False); y.backward(g)
def evalm():
    for _ in range(N):
        y = lsq(xs, ss, bs, 0, 127, 0, 255, eval_mode=True); y.backward(g)
def plain():
    for _ in range(N):
        y = xs * 2.0; y.backward(g)
for rep in range(3):
    for name, fn in (("direct", direct), ("dispatcher", disp), ("eval(masked)", evalm), ("plain mul autograd", plain)):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        print(rep, name, "%.1f us" % ((time.perf_counter() - t0) / N * 1e6))
