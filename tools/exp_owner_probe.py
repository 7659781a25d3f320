#!/usr/bin/env python3
"""Round-4 experiment (c): channel-OWNING workgroups for the per-channel backward of small-inner NCHW activations (BASELINE
config 5: [256,2048,7,7], bf16) -- no partials, no finalize launch -- probed with a no-arithmetic 2R:1W kernel of that access
pattern (tools/probes/owner_probe.hip) on COLD buffers, against ATen add and against the window pattern the product uses
(tools/probes/window_probe.hip: 4 KiB contiguous per row and workgroup).  us per launch; the kill criterion is the product's
own backward KERNEL time on this tensor (31 us, + 4.8 us finalize launch).
    python tools/exp_owner_probe.py            # table
    python tools/exp_owner_probe.py one U WAVES RUN XCD     # a few launches of one configuration (for rocprofv3 --pmc)"""
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


def main():
    own = build("owner_probe")
    win = build("window_probe")
    own.owner_probe_run.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 2 + [ctypes.c_void_p]
    win.window_probe_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    rows, C, inner = 256, 2048, 49
    L = C * inner                      # bf16 elements per row
    rp = L * 2 // 16                   # 16-byte packets per row
    n_f32 = rows * L // 2              # the buffers as fp32 words
    K = 12
    xs = [torch.randn(n_f32, device=dev) for _ in range(K)]
    gs = [torch.randn(n_f32, device=dev) for _ in range(K)]
    y = torch.empty(n_f32, device=dev)
    nbytes = 3 * rows * L * 2

    if len(sys.argv) > 1 and sys.argv[1] == "one":
        u, waves, run, xcd = (int(v) for v in sys.argv[2:6])
        for k in range(2 * K):
            assert own.owner_probe_run(u, waves, run, xcd, xs[k % K].data_ptr(), gs[k % K].data_ptr(), y.data_ptr(), rows, rp, None) == 0
        torch.cuda.synchronize()
        return

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

    print("# tools/exp_owner_probe.py: [256,2048,7,7] bf16 (%d MB per launch at 6 B/element), cold (x, grad rotate through %d sets); us per launch"
          % (nbytes >> 20, K))
    t = timeit([(lambda s, k=k: torch.add(gs[k], xs[k], out=y)) for k in range(K)], 2 * K)
    print("ATen add (fp32 view of the same bytes)            %6.1f us  %.2f TB/s" % (t, nbytes / t / 1e6))
    for p, u, splits in ((1, 4, 15), (1, 4, 20)):
        t = timeit([(lambda s, k=k: win.window_probe_run(p, u, xs[k].data_ptr(), gs[k].data_ptr(), y.data_ptr(), rows, L // 2, splits, s))
                    for k in range(K)], 2 * K)
        print("window pattern (product): 4 KiB x %2d row slabs     %6.1f us  %.2f TB/s" % (splits, t, nbytes / t / 1e6))
    # run = packets per row and owner: 49 = 8 channels (784 B, 16-byte aligned), 98 = 16 channels, 196 = 32, 392 = 64 channels
    # (6272 B = 49 whole 128-byte lines: no shared lines at all, but only 32 owners)
    for run in (49, 98, 196, 392):
        for waves in (4, 8, 16):
            for u in (1, 2, 4):
                for xcd in (0, 1):
                    if (rp // run) % 8 and xcd:
                        continue
                    t = timeit([(lambda s, k=k: own.owner_probe_run(u, waves, run, xcd, xs[k].data_ptr(), gs[k].data_ptr(), y.data_ptr(), rows, rp, s))
                                for k in range(K)], 2 * K)
                    print("owners of %3d packets/row (%2d channels, %4d owners) x %2d waves, U %d, xcd-aware %d   %6.1f us  %.2f TB/s"
                          % (run, run * 16 // 98, rp // run, waves, u, xcd, t, nbytes / t / 1e6), flush=True)
    torch.testing.assert_close(y, gs[(2 * K - 1) % K] + xs[(2 * K - 1) % K])


if __name__ == "__main__":
    main()
