#!/usr/bin/env python3
"""Row-group-window backward (quantized axis last): ring copies with / without the streaming hint, default policy otherwise,
(a) gradient fresh from a producer kernel + x cold, (b) both cold.  us per backward (graph differencing for (a))."""
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
dev = torch.device("cuda:0")


def graph_time(body, reps):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        body(0)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for k in range(reps):
                body(k)
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


for shape, axis in (((8192, 4096), 1), ((65536, 1024), 1), ((64, 56, 56, 256), 3), ((256, 197, 768), 2), ((64, 197, 768), 2), ((16384, 2048), 1)):
    for dt in (torch.bfloat16, torch.float32):
        n = 1
        for d in shape: n *= d
        esz = 2 if dt == torch.bfloat16 else 4
        K = max(3, min(16, -(-(1100 << 20) // (n * esz * 3))))
        a_ = [synth.normal_like(n, 1 + k, 0.5, 1.0, device=dev, dtype=dt).view(shape) for k in range(K)]
        b_ = [synth.normal_like(n, 100 + k, 0.0, 1e-3, device=dev, dtype=dt).view(shape) for k in range(K)]
        g_ = [torch.empty_like(a_[0]) for _ in range(K)]
        C = shape[axis]
        s = synth.uniform_like(C, 3, 0.02, 0.05, device=dev); b = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        q = (0, 127, 0, 255, True, 1.0, False, False, False)
        for k in range(K):
            torch.add(a_[k], b_[k], out=g_[k])
        t_prod = graph_time(lambda k: torch.add(a_[k % K], b_[k % K], out=g_[k % K]), K)
        res = []
        for knob in (2, 1, 2, 1):
            lib.lsq_hip_debug_set_ring_nt(knob)
            E._WS_BYTES_PC.clear()
            def both(k):
                torch.add(a_[k % K], b_[k % K], out=g_[k % K])
                E.hip_backward_per_channel(g_[k % K], a_[(k + K // 2) % K], s, b, axis, *q)
            ta = graph_time(both, K) - t_prod
            tc = graph_time(lambda k: E.hip_backward_per_channel(g_[k % K], a_[(k + K // 2) % K], s, b, axis, *q), K)
            res.append("%s: fresh grad %.1f, cold %.1f" % ("nt" if knob == 1 else "plain", ta, tc))
        lib.lsq_hip_debug_set_ring_nt(0)
        print("%-9s %-18s x%d  %s" % (str(dt).replace("torch.", ""), shape, K, " | ".join(res)), flush=True)
        del a_, b_, g_
