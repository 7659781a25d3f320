#!/usr/bin/env python3
"""Sweep the launch variants of the fp32 per-tensor kernels on the MI355X (tuning tool, not product).

    python tools/tune_stream.py --build         # here: hipcc -DLSQ_TUNING -> tools/_tune/liblsq_hip_tune.so
    gpurun -- python tools/tune_stream.py       # on the GPU box: sweep, print a table, write gpurun_out/tune.json

Variants: unroll {1,2,4,8} x nt-load {0,1} x nt-store {0,1} x workgroups-per-CU {2..16}; timing =
median of HIP-event times over `--iters` back-to-back launches, interleaved rounds (all variants per
round) so DVFS / thermal drift hits every variant alike.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lsqfakequantize-pytorch_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_tune")
SO = os.path.join(OUT, "liblsq_hip_tune.so")
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))


def build():
    os.makedirs(OUT, exist_ok=True)
    flags = ["-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
             "-DLSQ_TOOLS", "-DLSQ_TUNING", "-shared"]
    srcs = [os.path.join(CSRC, f) for f in ("lsq_capi.hip", "lsq_per_tensor.hip", "lsq_per_channel.hip", "lsq_observe.hip")]
    srcs.append(os.path.join(ROOT, "tools", "stream_probe.hip"))
    cmd = ["/opt/rocm/bin/hipcc"] + flags + srcs + ["-o", SO]
    print(" ".join(cmd))
    subprocess.check_call(cmd)


def enc(unroll, ntl, nts, bpc, chunked=0):
    return unroll | (int(ntl) << 8) | (int(nts) << 9) | (int(chunked) << 10) | (bpc << 16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--shape", default="128,512,56,56")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--probe", action="store_true", help="also measure no-arithmetic HBM ceilings (copy / add / read)")
    ap.add_argument("--calibrate", action="store_true", help="run ONLY 3 launches each of copy/add/read probes + product "
                    "fwd/bwd (for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE against known byte counts)")
    a = ap.parse_args()
    if a.build:
        build()
        return
    import torch
    from torchlsq import synth
    from torchlsq.extension import LsqParams
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lsq_tools
    C_ABI_INTERNAL = lsq_tools.internal_abi()
    lib = ctypes.CDLL(SO)
    for name, (res, args) in C_ABI_INTERNAL.items():
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args
    dev = torch.device("cuda:0")
    shape = tuple(int(v) for v in a.shape.split(","))
    x, g, scale, shift = synth.make_inputs("cfg2", device=dev, dtype=torch.float32, shape=shape)
    n = x.numel()
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    ds = torch.empty(1, device=dev)
    db = torch.empty(1, device=dev)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    p = LsqParams(0, 127, 0, 255, 1, 0, 0, 0, 1.0, 0)
    stream = torch.cuda.current_stream().cuda_stream
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)

    def fwd(v):
        rc = lib.lsq_hip_forward_per_tensor_ex(0, x.data_ptr(), y.data_ptr(), n, scale.data_ptr(), shift.data_ptr(),
                                               ctypes.byref(p), None, stream, v)
        assert rc == 0

    def bwd(v):
        rc = lib.lsq_hip_backward_per_tensor_ex(0, g.data_ptr(), x.data_ptr(), dx.data_ptr(), ds.data_ptr(), db.data_ptr(),
                                                None, n, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None,
                                                ws.data_ptr(), ws.numel(), stream, v)
        assert rc == 0

    if a.calibrate:
        lib.lsq_probe_run.restype = ctypes.c_int
        lib.lsq_probe_run.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int,
                                                                                    ctypes.c_void_p, ctypes.c_void_p]
        sink = torch.zeros(4, device=dev)
        for kind in (0, 1, 2):
            for _ in range(3):
                assert lib.lsq_probe_run(kind, 4, 256, x.data_ptr(), g.data_ptr(), y.data_ptr(), n, 512, sink.data_ptr(), stream) == 0
        for _ in range(3):
            fwd(0)
            bwd(0)
        torch.cuda.synchronize()
        print("calibration launches done: n = %d elements (%d bytes per fp32 tensor)" % (n, 4 * n))
        return
    unrolls = (2, 4, 8) if a.quick else (1, 2, 4, 8)
    bpcs = (1, 2, 4, 8, 16) if a.quick else (1, 2, 3, 4, 6, 8, 12, 16)
    variants = [(u, l, s, b, c) for u in unrolls for (l, s) in ((1, 1), (0, 1), (0, 0)) for b in bpcs for c in (0, 1)]
    if a.probe:
        lib.lsq_probe_run.restype = ctypes.c_int
        lib.lsq_probe_run.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int,
                                                                                    ctypes.c_void_p, ctypes.c_void_p]
        sink = torch.zeros(4, device=dev)
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        print("== HBM ceilings for the traffic shapes (no arithmetic), %d CUs" % cus)
        rows = []
        for kind, name, nbytes in ((0, "copy 1R:1W", 8 * n), (1, "add 2R:1W", 12 * n), (2, "read 2R:0W", 8 * n)):
            for block in (256, 512, 1024):
                for unroll in (2, 4, 8):
                    for wg_per_cu in (1, 2, 4, 8, 16):
                        if wg_per_cu * block > 2048:
                            continue
                        grid = cus * wg_per_cu
                        ts = []
                        for _ in range(3):
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            for _ in range(a.iters):
                                rc = lib.lsq_probe_run(kind, unroll, block, x.data_ptr(), g.data_ptr(), y.data_ptr(), n, grid,
                                                       sink.data_ptr(), stream)
                                assert rc == 0
                            e1.record()
                            e1.synchronize()
                            ts.append(e0.elapsed_time(e1) / a.iters)
                        med = sorted(ts)[1]
                        rows.append(dict(kind=name, block=block, unroll=unroll, wg_per_cu=wg_per_cu, ms=round(med, 5),
                                         gbs=round(nbytes / med / 1e6, 1)))
        for name in ("copy 1R:1W", "add 2R:1W", "read 2R:0W"):
            best = sorted([r for r in rows if r["kind"] == name], key=lambda r: -r["gbs"])
            for r in best[:5]:
                print("   %-11s block=%-4d unroll=%d wg/cu=%-2d  %.4f ms %8.1f GB/s" % (name, r["block"], r["unroll"], r["wg_per_cu"], r["ms"], r["gbs"]))
            print("   %-11s worst: %s" % (name, best[-1]))
        with open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w") as f:
            json.dump(rows, f, indent=1)
    results = {}
    for kind, fn, nbytes in (("fwd", fwd, 8 * n), ("bwd", bwd, 12 * n)):
        times = {v: [] for v in variants}
        for v in variants:       # warm every code object
            fn(enc(*v))
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for v in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                code = enc(*v)
                e0.record()
                for _ in range(a.iters):
                    fn(code)
                e1.record()
                e1.synchronize()
                times[v].append(e0.elapsed_time(e1) / a.iters)
        table = []
        for v in variants:
            t = sorted(times[v])
            med = t[len(t) // 2]
            table.append(dict(unroll=v[0], nt_load=v[1], nt_store=v[2], bpc=v[3], chunked=v[4], ms_med=round(med, 5), ms_min=round(t[0], 5),
                              gbs_med=round(nbytes / med / 1e6, 1), gbs_best=round(nbytes / t[0] / 1e6, 1)))
        table.sort(key=lambda r: r["ms_med"])
        results[kind] = table
        print("== %s (%d elements, %.0f MB algorithmic per launch)" % (kind, n, nbytes / 1e6))
        for r in table[:12]:
            print("   unroll=%d ntl=%d nts=%d bpc=%-2d chunked=%d  med %.4f ms  %7.1f GB/s  (best %7.1f)" %
                  (r["unroll"], r["nt_load"], r["nt_store"], r["bpc"], r["chunked"], r["ms_med"], r["gbs_med"], r["gbs_best"]))
        print("   ...")
        for r in table[-4:]:
            print("   unroll=%d ntl=%d nts=%d bpc=%-2d chunked=%d  med %.4f ms  %7.1f GB/s" %
                  (r["unroll"], r["nt_load"], r["nt_store"], r["bpc"], r["chunked"], r["ms_med"], r["gbs_med"]))
    with open(os.path.join(ROOT, "gpurun_out", "tune.json"), "w") as f:
        json.dump(dict(shape=shape, iters=a.iters, rounds=a.rounds, results=results), f, indent=1)


if __name__ == "__main__":
    main()
