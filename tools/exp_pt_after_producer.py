#!/usr/bin/env python3
"""Per-tensor K1 / K2 right after a producer kernel wrote their input (grad for K2, x for K1), and cold: non-temporal loads
(shipped) against plain loads, via the tuning build (tools/_tune/liblsq_hip_tune.so).  us per launch (graph differencing)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from torchlsq import synth
from torchlsq.extension import C_ABI, LsqParams
import lsq_tools
C_ABI_INTERNAL = lsq_tools.internal_abi()
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_tune", "liblsq_hip_tune.so"))
for tbl in (C_ABI, C_ABI_INTERNAL):
    for name, (res, args) in tbl.items():
        try:
            getattr(lib, name).restype = res; getattr(lib, name).argtypes = args
        except AttributeError:
            pass
dev = torch.device("cuda:0")


def graph_time(body, reps):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        s = st.cuda_stream
        body(0, s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for k in range(reps):
                body(k, s)
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


for n in (6422528, 25690112, 51380224):
    K = max(3, min(16, -(-(1100 << 20) // (n * 4 * 3))))
    a_ = [synth.normal_like(n, 1 + k, 0.5, 1.0, device=dev) for k in range(K)]
    b_ = [synth.normal_like(n, 100 + k, 0.0, 1e-3, device=dev) for k in range(K)]
    t_ = [torch.empty(n, device=dev) for _ in range(K)]      # the producer's output
    out = torch.empty(n, device=dev)
    scale = torch.tensor([0.03], device=dev); shift = torch.tensor([0.1], device=dev)
    ds = torch.empty(1, device=dev); db = torch.empty(1, device=dev); ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    p = LsqParams(0, 127, 0, 255, 1, 0, 0, 0, 1.0, 0)
    prod = lambda k, s: torch.add(a_[k % K], b_[k % K], out=t_[k % K])
    t_prod = graph_time(prod, K)
    res = []
    for name, vb, vf in (("nt loads (shipped)", 4 | (3 << 8) | (2 << 16), 4 | (3 << 8) | (16 << 16)), ("plain loads", 4 | (2 << 8) | (2 << 16), 4 | (2 << 8) | (16 << 16)),
                         ("plain loads+stores", 4 | (2 << 16), 4 | (16 << 16))):
        def bwd(k, s):   # grad = producer output (fresh), x cold
            assert lib.lsq_hip_backward_per_tensor_ex(0, t_[k % K].data_ptr(), a_[(k + K // 2) % K].data_ptr(), out.data_ptr(), ds.data_ptr(), db.data_ptr(), None,
                                                      n, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None, ws.data_ptr(), ws.numel(), s, vb) == 0
        def fwd(k, s):   # x = producer output (fresh)
            assert lib.lsq_hip_forward_per_tensor_ex(0, t_[k % K].data_ptr(), out.data_ptr(), n, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None, s, vf) == 0
        tb = graph_time(lambda k, s: (prod(k, s), bwd(k, s)), K) - t_prod
        tbc = graph_time(bwd, K)
        tf = graph_time(lambda k, s: (prod(k, s), fwd(k, s)), K) - t_prod
        tfc = graph_time(fwd, K)
        res.append("%s: bwd after producer %.1f cold %.1f | fwd after producer %.1f cold %.1f" % (name, tb, tbc, tf, tfc))
    print("n=%d x%d  producer %.1f || %s" % (n, K, t_prod, "  ||  ".join(res)), flush=True)
    del a_, b_, t_
