#!/usr/bin/env python3
"""Depth of the backward's LDS-DMA ring on COLD buffers: 4 stages (shipped) against 2 and 3 (experiment builds
tools/_tune/liblsq_hip_depth{2,3}.so, -DLSQ_BWD_DMA_DEPTH=N) -- a shallower ring takes less LDS, so more workgroups fit a
CU (the 16-bit backward is short of waves, not of bytes in flight).  Ring forced, N workgroups per CU; us per backward."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torchlsq  # noqa: F401
from torchlsq import extension as E, synth
from torchlsq.extension import C_ABI, LsqParams
import lsq_tools
C_ABI_INTERNAL = lsq_tools.internal_abi()

dev = torch.device("cuda:0")
libs = {"4": E.library()}
for d in ("2", "3"):
    libs[d] = ctypes.CDLL(os.path.join(ROOT, "tools", "_tune", "liblsq_hip_depth%s.so" % d))
for lib in libs.values():
    for tbl in (C_ABI, C_ABI_INTERNAL):
        for name, (res, args) in tbl.items():
            getattr(lib, name).restype = res
            getattr(lib, name).argtypes = args
RING = 2 << 12


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


for (outer, C, inner) in ((256, 2048, 49), (512, 2048, 49), (128, 1024, 49), (64, 512, 196), (256, 512, 49), (128, 512, 784), (32, 256, 3136)):
    for dt, code in ((torch.bfloat16, 2),):
        n = outer * C * inner
        esz = 2 if code == 2 else 4
        K = max(2, min(16, -(-(1100 << 20) // (n * esz * 2))))
        xs = [synth.normal_like(n, 1 + k, 0.5, 1.0, device=dev, dtype=dt) for k in range(K)]
        gs = [synth.normal_like(n, 100 + k, 0.0, 1e-3, device=dev, dtype=dt) for k in range(K)]
        scale = synth.uniform_like(C, 3, 0.02, 0.05, device=dev); shift = synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        p = LsqParams(0, 127, 0, 255, 1, 0, 0, 0, 1.0, 0)
        ws = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
        dx = torch.empty_like(xs[0]); ds = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
        out = []
        for rnd in range(3):                       # interleaved order, three rounds: the first measurement of a run is slow
            for name in ("4", "2", "3"):
                lib = libs[name]
                for bpc in (4, 6):
                    v = 1 | (3 << 8) | (bpc << 16) | RING
                    def mk(k, lib=lib, v=v):
                        def bwd(s):
                            assert lib.lsq_hip_backward_per_channel_ex(code, gs[k].data_ptr(), xs[k].data_ptr(), dx.data_ptr(), ds.data_ptr(), db.data_ptr(), None,
                                                                       outer, C, inner, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None,
                                                                       ws.data_ptr(), ws.numel(), s, v) == 0
                        return bwd
                    out.append("d%s/%d %.1f" % (name, bpc, timeit([mk(k) for k in range(K)], 2 * K)))
            out.append("|")
        print("%-8s [%d,%d,%d] x%d bwd us: %s" % (str(dt).replace("torch.", ""), outer, C, inner, K, "  ".join(out)), flush=True)
        del xs, gs
