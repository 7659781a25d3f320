#!/usr/bin/env python3
"""Store cache policy x load policy for the forward (1R:1W) and backward (2R:1W) traffic shapes at config-2 size: a HIP graph of
alternating copy / add kernels (so kernel boundaries count), per-'step' time."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_tune", "libpolicy_probe.so"))
lib.policy_probe_run.restype = ctypes.c_int
lib.policy_probe_run.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
n = 128 * 512 * 56 * 56
x = torch.randn(n, device=dev); g = torch.randn(n, device=dev); y = torch.empty(n, device=dev); dx = torch.empty(n, device=dev)
names = {0: "plain", 1: "nt", 2: "sc0 sc1", 3: "sc1", 4: "sc0 sc1 nt", 5: "sc0"}


def timeit(fn, reps=10):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        s = st.cuda_stream
        fn(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps):
                fn(s)
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


for ld in (1, 0):
    for pol in range(6):
        for gf, gb in ((4096, 512),):
            def step(s):
                assert lib.policy_probe_run(0, pol, ld, x.data_ptr(), g.data_ptr(), y.data_ptr(), n, gf, s) == 0
                assert lib.policy_probe_run(1, pol, ld, x.data_ptr(), g.data_ptr(), dx.data_ptr(), n, gb, s) == 0
            def fwd(s):
                assert lib.policy_probe_run(0, pol, ld, x.data_ptr(), g.data_ptr(), y.data_ptr(), n, gf, s) == 0
            def bwd(s):
                assert lib.policy_probe_run(1, pol, ld, x.data_ptr(), g.data_ptr(), dx.data_ptr(), n, gb, s) == 0
            t, tf, tb = timeit(step), timeit(fwd), timeit(bwd)
            print("loads %-5s stores %-10s  copy %.1f us (%.0f GB/s)  add %.1f us (%.0f GB/s)  copy+add step %.1f us = %.1f GElem/s-equivalent" %
                  ("nt" if ld else "plain", names[pol], tf, 8 * n / tf / 1e3, tb, 12 * n / tb / 1e3, t, n / t / 1e3))
