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
    for src, extra in (("lsq_capi.hip", []), ("lsq_per_tensor.hip", []), ("lsq_observe.hip", []), ("lsq_multi.hip", []), ("lsq_comm.hip", []),
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
    ap.add_argument("--own", action="store_true", help="config 5 (bf16, fp32) and a small tensor as OWNER windows (lsq_hip_debug_set_own(1)) next to the 256-lane windows")
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
             ("[32,256,56,56] bf16 axis 1", (32, 256, 56, 56), 1, torch.bfloat16, (-8, 7, -128, 127)),
             ("vit bf16 [64,197,768] axis 2", (64, 197, 768), 2, torch.bfloat16, (0, 127, 0, 255)),
             ("vit fp32 [64,197,768] axis 2", (64, 197, 768), 2, torch.float32, (0, 127, 0, 255))]
    if a.own:
        base_cases = [("cfg5 bf16 [256,2048,7,7] axis 1", (256, 2048, 7, 7), 1, torch.bfloat16, (-8, 7, -128, 127)),
                      ("cfg5 fp32 [256,2048,7,7] axis 1", (256, 2048, 7, 7), 1, torch.float32, (-8, 7, -128, 127)),
                      ("[64,2048,7,7] bf16 axis 1", (64, 2048, 7, 7), 1, torch.bfloat16, (-8, 7, -128, 127))]
        cases = []
        for c_ in base_cases:
            cases.append(("WINDOWS " + c_[0],) + c_[1:] + (2,))
            cases.append(("OWNERS  " + c_[0],) + c_[1:] + (1,))
            cases.append(("OWNERS without the priority turns " + c_[0],) + c_[1:] + (3,))
    else:
        cases = [c_ + (0,) for c_ in cases]
    print("# tools/exp_timeline.py: per-wave shader-clock stamps of the window-mode backward (one launch, cold inputs); us")
    for name, shape, axis, dtype, q, own in cases:
        lsq_tools.set_knob("set_own", own)
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
        # (the kernel numbers its waves with its LAUNCH BOUND's waves per workgroup: 8 for owner windows whatever they launch)
        per_wg = 8 if note["kind"] == "owners" else note["block"] // 64
        waves = note["grid_x"] * note["grid_y"] * per_wg
        rec = buf[:waves * 8].view(waves, 8).cpu().numpy().astype(np.float64)
        wave_in_wg = np.arange(waves) % per_wg
        wave_in_wg = wave_in_wg[rec[:, 0] > 0]
        rec = rec[rec[:, 0] > 0]
        # every XCD counts its own shader clock: stamps are comparable inside one XCC only, so entry / exit times are taken
        # relative to the first entry on the same XCC; the HW_ID register (bits 8-11 CU, 13-15 SE on gfx9) gives the CU
        xcc = rec[:, 7].astype(np.int64) & 0xf
        hw = rec[:, 6].astype(np.int64)
        cu_key = xcc * 65536 + ((hw >> 8) & 0xff)            # HW_ID: CU id bits 8-11, SH id bit 12, SE id bits 13-15
        base = np.zeros(len(rec)); last = np.zeros(len(rec))
        per_cu = []
        for k in np.unique(cu_key):
            m = cu_key == k
            base[m] = rec[m, 0].min()
            last[m] = rec[m, 3].max()
            per_cu.append((last[m][0] - base[m][0], int(m.sum())))
        xcc = cu_key                                             # (the statistics below are per CU)
        op_us = e0.elapsed_time(e1) * 1e3
        print("## %s: grid %d x %d x %d lanes, ring depth %d, nt %d, %d resident per CU assumed, %d waves stamped; this op (kernel + finalize launch) %.1f us" % (
            name, note["grid_x"], note["grid_y"], note["block"], note["ring_depth"], note["ring_nt"], note["resident_per_cu"], len(rec), op_us))
        for label, ghz in (("at 2.1 GHz", 2100.0),):
            us = lambda c: c / ghz
            q_ = lambda v: "min %6.2f  p10 %6.2f  p25 %6.2f  median %6.2f  p75 %6.2f  p90 %6.2f  max %6.2f" % tuple(us(np.percentile(v, p)) for p in (0, 10, 25, 50, 75, 90, 100))
            spans = np.array([p_[0] for p_ in per_cu])
            print("  [%s] %d CUs seen; busy span per CU (its first wave's entry -> its last wave's exit): " % (label, len(per_cu)) + q_(spans) +
                  ";  waves per CU min %d max %d" % (min(p_[1] for p_ in per_cu), max(p_[1] for p_ in per_cu)))
            print("    wave entry after its CU's first   " + q_(rec[:, 0] - base))
            print("    prologue (entry -> row loop)      " + q_(rec[:, 1] - rec[:, 0]))
            print("    row loop                          " + q_(rec[:, 2] - rec[:, 1]))
            print("      of which in the ring's waits    " + q_(rec[:, 4]))
            print("      per row                         " + q_((rec[:, 2] - rec[:, 1]) / np.maximum(rec[:, 5], 1)))
            print("    epilogue (row loop -> exit)       " + q_(rec[:, 3] - rec[:, 2]))
            print("    wave life (entry -> exit)         " + q_(rec[:, 3] - rec[:, 0]))
            print("    wave exit before its CU's last    " + q_(last - rec[:, 3]))
            # how the waves of a CU spread over its four SIMDs (HW_ID bits 4-5), and what a wave's row costs by the company it keeps
            simd = (hw >> 4) & 3
            simd_key = cu_key * 4 + simd
            uniq, inv, cnt = np.unique(simd_key, return_inverse=True, return_counts=True)
            per_row = us((rec[:, 2] - rec[:, 1]) / np.maximum(rec[:, 5], 1))
            print("    waves per SIMD over the launch: " + " ".join("%d:%d" % (v, int((cnt == v).sum())) for v in np.unique(cnt)) +
                  "   (SIMDs seen: %d of %d)" % (len(uniq), 4 * len(per_cu)))
            # concurrency: waves of the same SIMD alive at this wave's mid-loop
            mid = (rec[:, 1] + rec[:, 2]) / 2
            conc = np.zeros(len(rec), dtype=np.int64)
            order = np.argsort(simd_key, kind="stable")
            start = 0
            sk = simd_key[order]
            for end in list(np.nonzero(np.diff(sk))[0] + 1) + [len(sk)]:
                idx = order[start:end]
                for i in idx:
                    conc[i] = int(((rec[idx, 0] <= mid[i]) & (rec[idx, 3] > mid[i])).sum())
                start = end
            for c_ in np.unique(conc):
                m_ = conc == c_
                print("      waves with %d wave(s) alive on their SIMD at mid-loop: %5d   row time median %.2f us  (p10 %.2f, p90 %.2f)" % (
                    c_, int(m_.sum()), np.median(per_row[m_]), np.percentile(per_row[m_], 10), np.percentile(per_row[m_], 90)))
            if note["kind"] == "owners":     # which waves of a workgroup are the slow ones?
                loop_us = us(rec[:, 2] - rec[:, 1])
                for w in np.unique(wave_in_wg):
                    m_ = wave_in_wg == w
                    print("      wave %d of its workgroup: %5d   rows %3d   row loop median %6.2f us (p10 %6.2f, p90 %6.2f)   SIMD %s" % (
                        w, int(m_.sum()), int(np.median(rec[m_, 5])), np.median(loop_us[m_]), np.percentile(loop_us[m_], 10),
                        np.percentile(loop_us[m_], 90), " ".join("%d:%d" % (v, int((simd[m_] == v).sum())) for v in range(4))))
                xcd_of = rec[:, 7].astype(np.int64) & 0xf
                for x_ in np.unique(xcd_of):
                    m_ = xcd_of == x_
                    print("      XCC %d: %5d waves   row loop median %6.2f us (p10 %6.2f, p90 %6.2f)" % (
                        x_, int(m_.sum()), np.median(loop_us[m_]), np.percentile(loop_us[m_], 10), np.percentile(loop_us[m_], 90)))
            # waves alive over time on one CU (the one with the most waves)
            kbig = max(np.unique(cu_key), key=lambda k: int((cu_key == k).sum()))
            m = cu_key == kbig
            t0s, t3s = us(rec[m, 0] - base[m]), us(rec[m, 3] - base[m])
            grid_t = np.linspace(0, t3s.max(), 16)
            alive = [int(((t0s <= t) & (t3s > t)).sum()) for t in grid_t]
            print("    waves alive on one CU at t = " + " ".join("%.1f:%d" % (t, a_) for t, a_ in zip(grid_t, alive)))
        sys.stdout.flush()
        del xs, gs, buf
        torch.cuda.empty_cache()
    lsq_tools.deactivate()


if __name__ == "__main__":
    main()
