#!/usr/bin/env python3
"""Window-pattern 2R:1W probe (tools/probes/window_probe.hip) on COLD buffers: contiguous bytes per row and workgroup
(P x 4 KiB) x rows in flight (U) x workgroups, against ATen add on the same buffers.  us per launch and TB/s."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_tune", "libwindow_probe.so"))
lib.window_probe_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                 ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")


def timeit(fns, reps):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        s = st.cuda_stream
        for f in fns: f(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for k in range(reps):
                fns[k % len(fns)](s)
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


for rows, L in ((256, 2048 * 49 // 2), (256, 2048 * 49), (128, 512 * 784), (8192, 4096), (12608, 768)):     # (the first: config 5 in bf16, as fp32 words)
    n = rows * L
    K = max(2, min(12, -(-(1100 << 20) // (n * 8))))
    xs = [torch.randn(n, device=dev) for _ in range(K)]
    gs = [torch.randn(n, device=dev) for _ in range(K)]
    y = torch.empty(n, device=dev)
    out = []
    t = timeit([(lambda s, k=k: torch.add(gs[k], xs[k], out=y)) for k in range(K)], 2 * K)
    out.append("ATen add %.1f (%.2f TB/s)" % (t, 12 * n / t / 1e6))
    for p, u in ((1, 4), (1, 8), (2, 2), (2, 4), (4, 1), (4, 2), (8, 1)):
        n_win = -(-(L // 4) // (256 * p))
        for wgs in (512, 1024, 2048, 4096):
            splits = max(1, min(rows, wgs // n_win))
            def mk(k, p=p, u=u, splits=splits):
                def f(s):
                    assert lib.window_probe_run(p, u, xs[k].data_ptr(), gs[k].data_ptr(), y.data_ptr(), rows, L, splits, s) == 0
                return f
            t = timeit([mk(k) for k in range(K)], 2 * K)
            out.append("P%d U%d %dx%d %.1f" % (p, u, n_win, splits, t))
    torch.testing.assert_close(y, gs[(2 * K - 1) % K] + xs[(2 * K - 1) % K])
    print("[%d,%d] x%d (8 TB/s = %.1f us): %s" % (rows, L, K, 12 * n / 8e6, "  ".join(out)), flush=True)
    del xs, gs, y
