#!/usr/bin/env python3
"""Where does a window-mode backward workgroup spend its life?  (experiment build, -DLSQ_TOOLS -DLSQ_TIMELINE)

    python tools/exp_timeline.py --build      # here: hipcc -> tools/_tune/liblsq_hip_timeline.so  (~3 min)
    python tools/exp_timeline.py              # on the GPU box

Every wave of bwd_pc_kernel stamps the shader clock (s_memtime) at entry, before its row loop, after it and at its end, and
adds up the cycles it sat in the ring's s_waitcnt; the table below is the distribution of those intervals over all waves of
one launch on cold inputs, in microseconds (cycles / the clock implied by the launch's HIP-event duration), next to the
launch's wall time.  Output: profiles/r03_k4_timeline.txt."""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lsqfakequantize-pytorch_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_tune")
SO = os.path.join(OUT, "liblsq_hip_timeline.so")
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def build():
    os.makedirs(OUT, exist_ok=True)
    flags = ["-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-DLSQ_TOOLS", "-DLSQ_TIMELINE"]
    jobs, objs = [], []
    for src, extra in (("lsq_capi.hip", []), ("lsq_per_tensor.hip", []), ("lsq_observe.hip", []), ("lsq_multi.hip", []),
                       ("lsq_per_channel.hip", ["-DLSQ_PC_IO=io_f32"]), ("lsq_per_channel.hip", ["-DLSQ_PC_IO=io_f64"]),
                       ("lsq_per_channel.hip", ["-DLSQ_PC_IO=io_bf16"]), ("lsq_per_channel.hip", ["-DLSQ_PC_IO=io_f16"])):
        obj = os.path.join("/tmp", "tl_%s_%s.o" % (src.replace(".hip", ""), extra[0][-4:] if extra else "x"))
        objs.append(obj)
        jobs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + flags + extra + ["-c", os.path.join(CSRC, src), "-o", obj]))
    for j in jobs:
        assert j.wait() == 0
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950"] + objs + ["-o", SO])
    print("built", SO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    a = ap.parse_args()
    if a.build:
        build()
        return
    import numpy as np
    import torch
    import torchlsq  # noqa: F401
    import lsq_tools
    from torchlsq import extension as E, synth
    lib = lsq_tools.activate(SO)
    lib.lsq_hip_debug_set_timeline.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    cases = [("cfg5 bf16 [256,2048,7,7] axis 1", (256, 2048, 7, 7), 1, torch.bfloat16, (-8, 7, -128, 127)),
             ("cfg5 fp32 [256,2048,7,7] axis 1", (256, 2048, 7, 7), 1, torch.float32, (-8, 7, -128, 127)),
             ("tok bf16 [8192,4096] axis 1", (8192, 4096), 1, torch.bfloat16, (0, 127, 0, 255)),
             ("[32,256,56,56] bf16 axis 1", (32, 256, 56, 56), 1, torch.bfloat16, (-8, 7, -128, 127))]
    print("# tools/exp_timeline.py: per-wave shader-clock stamps of the window-mode backward (one launch, cold inputs); us")
    for name, shape, axis, dtype, q in cases:
        n = int(np.prod(shape))
        esz = 2 if dtype == torch.bfloat16 else 4
        K = max(2, -(-(600 << 20) // (2 * n * esz)))
        xs = [synth.normal_like(n, 10 + k, 0.0, 1.0, dtype=dtype, device=dev).view(shape) for k in range(K)]
        gs = [synth.normal_like(n, 40 + k, 0.0, 1e-3, dtype=dtype, device=dev).view(shape) for k in range(K)]
        C = shape[axis]
        s, b = synth.uniform_like(C, 3, 0.05, 0.35, device=dev), synth.normal_like(C, 4, 0.0, 0.1, device=dev)
        args = q + (True, 1.0, False, False, False)
        buf = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
        lib.lsq_hip_debug_set_timeline(None)
        for k in range(K):
            E.hip_backward_per_channel(gs[k], xs[k], s, b, axis, *args)
        note = lsq_tools.last_launch()
        torch.cuda.synchronize()
        lib.lsq_hip_debug_set_timeline(buf.data_ptr())
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        E.hip_backward_per_channel(gs[0], xs[0], s, b, axis, *args)
        e1.record()
        torch.cuda.synchronize()
        lib.lsq_hip_debug_set_timeline(None)
        waves = note["grid_x"] * note["grid_y"] * (note["block"] // 64)
        rec = buf[:waves * 8].view(waves, 8).cpu().numpy().astype(np.float64)
        rec = rec[rec[:, 0] > 0]
        t_begin, t_end = rec[:, 0].min(), rec[:, 3].max()
        span = t_end - t_begin
        # the shader clock per us: the span of the stamps against the op's event time minus the finalize launch (~4.5 us + gap)
        op_us = e0.elapsed_time(e1) * 1e3
        print("## %s: grid %d x %d x %d lanes, ring depth %d, %d waves stamped; op (kernel + finalize launch) %.1f us; stamps span %.0f cycles" % (
            name, note["grid_x"], note["grid_y"], note["block"], note["ring_depth"], len(rec), op_us, span))
        for label, ghz in (("at 2.1 GHz", 2100.0), ("at 2.4 GHz", 2400.0)):
            us = lambda c: c / ghz
            q_ = lambda v: "min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % tuple(us(np.percentile(v, p)) for p in (0, 10, 50, 90, 100))
            print("  [%s] kernel span (first entry -> last exit) %.2f us" % (label, us(span)))
            print("    wave entry after first entry      " + q_(rec[:, 0] - t_begin))
            print("    prologue (entry -> row loop)      " + q_(rec[:, 1] - rec[:, 0]))
            print("    row loop                          " + q_(rec[:, 2] - rec[:, 1]))
            print("      of which in the ring's waits    " + q_(rec[:, 4]))
            print("      per row                         " + q_((rec[:, 2] - rec[:, 1]) / np.maximum(rec[:, 5], 1)))
            print("    epilogue (row loop -> exit)       " + q_(rec[:, 3] - rec[:, 2]))
            print("    wave exit before last exit        " + q_(t_end - rec[:, 3]))
        xcc = rec[:, 7].astype(np.int64) & 0xf
        print("    waves per XCC: %s" % np.bincount(xcc, minlength=8).tolist())
        sys.stdout.flush()
        del xs, gs, buf
        torch.cuda.empty_cache()
    lsq_tools.deactivate()


if __name__ == "__main__":
    main()
