#!/usr/bin/env python3
"""Round-5 experiment for the last-axis (token layout) per-channel backward: the COLUMN-BLOCK x ROW-SLAB access pattern
(tools/probes/colblock_probe.hip, no arithmetic, 2R:1W) against whole-row workgroups and ATen add, on the [64,197,768] bf16
activation (12608 rows of 96 packets) and [8192,4096] bf16 (512 packets per row), cold buffers.  us per launch.
    python tools/exp_colblock_probe.py"""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_tune")


def build(name):
    so = os.path.join(OUT, "lib%s.so" % name)
    src = os.path.join(ROOT, "tools", "probes", name + ".hip")
    if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(OUT, exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-shared", "-fPIC", "--offload-arch=gfx950", src, "-o", so])
    return ctypes.CDLL(so)


def timeit(fns, reps):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        s = st.cuda_stream
        for f in fns:
            f(s)
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


def main():
    lib = build("colblock_probe")
    lib.colblock_probe_run.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    for rows, rp, label in ((12608, 96, "[64,197,768] bf16"), (8192, 512, "[8192,4096] bf16"), (12608, 192, "[64,197,768] fp32")):
        n_f32 = rows * rp * 4
        K = max(4, min(24, (1 << 30) // (n_f32 * 8)))
        xs = [torch.randn(n_f32, device=dev) for _ in range(K)]
        gs = [torch.randn(n_f32, device=dev) for _ in range(K)]
        y = torch.empty(n_f32, device=dev)
        nbytes = 3 * n_f32 * 4
        print("# %s: %d rows x %d packets, %.1f MB per launch, cold (%d input sets); us per launch, TB/s" % (label, rows, rp, nbytes / 1e6, K))
        t = timeit([(lambda s, k=k: torch.add(gs[k], xs[k], out=y)) for k in range(K)], 2 * K)
        print("ATen add                                   %6.1f us  %.2f TB/s" % (t, nbytes / t / 1e6))
        widths = [w for w in (8, 16, 32, 64, 96, 128, 192, 256, 512) if rp % w == 0 and w <= 768]
        for w in widths:
            n_cb = rp // w
            for block in (256, 512, 768, 1024):
                if block < w or (block // w) * w < block * 3 // 4:
                    continue
                for wg_per_cu in (1, 2, 3, 4):
                    if block * wg_per_cu > 2048 or (block >= 768 and wg_per_cu > 2):
                        continue
                    slabs = max(1, (256 * wg_per_cu) // n_cb)
                    best = None
                    for u in (1, 2, 4):
                        t = timeit([(lambda s, k=k: lib.colblock_probe_run(u, block, w, slabs, xs[k].data_ptr(), gs[k].data_ptr(), y.data_ptr(),
                                                                          rows, rp, s)) for k in range(K)], 2 * K)
                        if best is None or t < best[0]:
                            best = (t, u)
                    print("w=%3d lanes (%4d B runs) x %2d column blocks, %4d-lane wg, %d/CU -> %4d slabs (%5.1f rows/lane)   %6.1f us  %.2f TB/s  (U=%d)"
                          % (w, 16 * w, n_cb, block, wg_per_cu, slabs, rows / slabs / (block // w), best[0], nbytes / best[0] / 1e6, best[1]), flush=True)
        del xs, gs, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
