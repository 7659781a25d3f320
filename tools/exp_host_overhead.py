#!/usr/bin/env python3
"""Where does the host time of one op call go? (diagnostic)"""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torch
import torchlsq
from torchlsq import synth, extension as E
from torchlsq.functional import lsq
dev = torch.device("cuda:0")
x, g, scale, shift = synth.make_inputs("cfg1", device=dev, dtype=torch.float32)
ops = torch.ops.torchlsq
q = (0, 127, 0, 255)
N = 3000
def loop_fwd_op():
    for _ in range(N): ops.lsq_forward_per_tensor(x, scale, shift, *q, True, 1.0, False, False, False)
def loop_bwd_op():
    for _ in range(N): ops.lsq_backward_per_tensor(g, x, scale, shift, *q, True, 1.0, False, False, False)
def loop_fwd_direct():
    for _ in range(N): E.hip_forward_per_tensor(x, scale, shift, *q, True, 1.0, False, False, False)
def loop_empty_like():
    for _ in range(N): torch.empty_like(x)
xs = x.clone().requires_grad_(True); ss = scale.clone().requires_grad_(True); bs = shift.clone().requires_grad_(True)
def loop_autograd():
    for _ in range(N):
        y = lsq(xs, ss, bs, 0, 127, 0, 255)
        y.backward(g)
for name, fn in (("fwd op", loop_fwd_op), ("bwd op", loop_bwd_op), ("fwd direct", loop_fwd_direct), ("empty_like", loop_empty_like), ("lsq()+backward", loop_autograd)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("%-16s %.2f us per call" % (name, t / N * 1e6))
pr = cProfile.Profile(); pr.enable(); loop_autograd(); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(16)
