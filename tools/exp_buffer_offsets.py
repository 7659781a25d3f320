#!/usr/bin/env python3
"""Does the relative placement of the input and output buffers matter?  Per-tensor kernels through the C ABI with explicit
pointers: x (and grad) at 2 MiB-aligned addresses, the output at a 2 MiB-aligned address PLUS an offset; GPU time per launch by
offset (HIP events around 20 launches, three rounds, median).  BASELINE config 2 (205 M fp32 elements) and a 25.7 M shard.
Output: profiles/r04_buffer_offsets.txt."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lsqfakequantize-pytorch_amd"))
import torchlsq  # noqa: F401,E402
from torchlsq import extension as E  # noqa: E402

dev = torch.device("cuda:0")
lib = E.library()
MB = 1 << 20


def aligned(nbytes, offset):
    """a byte tensor whose data_ptr is 2 MiB-aligned + offset"""
    raw = torch.empty(nbytes + 4 * MB + offset, dtype=torch.uint8, device=dev)
    base = (-raw.data_ptr()) % (2 * MB)
    return raw, raw.data_ptr() + base + offset


def time_launches(fn, reps=20):
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[2]


def main():
    print("# tools/exp_buffer_offsets.py: fp32 per-tensor kernels, us per launch; inputs 2 MiB-aligned, the output 2 MiB-aligned + offset")
    p = E.LsqParams(0, 127, 0, 255, 1, 0, 0, 0, 1.0, 0)
    scale = torch.tensor([0.03], device=dev); shift = torch.tensor([0.0], device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.lsq_hip_backward_per_tensor_workspace(0, 1 << 28), dtype=torch.uint8, device=dev)
    ds = torch.empty(1, device=dev); db = torch.empty(1, device=dev)
    for n in (205520896, 25690112):
        nb = 4 * n
        keep_x, px = aligned(nb, 0)
        keep_g, pg = aligned(nb, 0)
        xt = keep_x[(px - keep_x.data_ptr()):(px - keep_x.data_ptr()) + nb].view(torch.float32); xt.normal_(1.5, 1.0)
        gt = keep_g[(pg - keep_g.data_ptr()):(pg - keep_g.data_ptr()) + nb].view(torch.float32); gt.normal_(0.0, 1e-3)
        for off in (0, 256, 4096, 65536, 512 * 1024, MB, MB + 65536, 3 * MB // 2):
            keep_y, py = aligned(nb, off)
            fwd = lambda: lib.lsq_hip_forward_per_tensor(0, px, py, n, scale.data_ptr(), shift.data_ptr(), ctypes.byref(p), None, stream)
            bwd = lambda: lib.lsq_hip_backward_per_tensor(0, pg, px, py, ds.data_ptr(), db.data_ptr(), None, n, scale.data_ptr(), shift.data_ptr(),
                                                          ctypes.byref(p), None, ws.data_ptr(), ws.numel(), stream)
            assert fwd() == 0 and bwd() == 0
            torch.cuda.synchronize()
            tf = [time_launches(fwd) for _ in range(3)]
            tb = [time_launches(bwd) for _ in range(3)]
            print("%10d el  output offset %8d B   forward %s   backward %s" % (n, off, " ".join("%6.1f" % t for t in tf), " ".join("%6.1f" % t for t in tb)), flush=True)
            del keep_y
            torch.cuda.empty_cache()
        del keep_x, keep_g, xt, gt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
