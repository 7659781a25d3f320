#!/usr/bin/env python3
"""N per-channel weight quantizers: N single calls against ONE horizontally fused call (functional.lsq_foreach ->
lsq_hip_*_per_channel_multi), forward + backward.  Wall time per step (host-inclusive, eager, what a training loop pays)
and GPU time per step (the same steps replayed from a HIP graph).  Output: profiles/r03_foreach_weights.txt."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
import torchlsq  # noqa: E402,F401
from torchlsq import extension as E, synth  # noqa: E402
from torchlsq.functional import lsq, lsq_foreach  # noqa: E402

dev = torch.device("cuda:0")
KW = dict(quant_min=-128, quant_max=127, type_min=-128, type_max=127, is_affine=False)


def make(shapes, dtype):
    xs, gs, ss, bs = [], [], [], []
    for k, shape in enumerate(shapes):
        n = 1
        for d in shape:
            n *= d
        xs.append(synth.normal_like(n, 100 + k, 0.0, 0.05, dtype=dtype, device=dev).view(shape).requires_grad_(True))
        gs.append(synth.normal_like(n, 300 + k, 0.0, 1e-3, dtype=dtype, device=dev).view(shape))
        ss.append(synth.uniform_like(shape[0], 500 + k, 5e-4, 2.5e-3, device=dev).requires_grad_(True))
        bs.append(torch.zeros(shape[0], device=dev).requires_grad_(True))
    return xs, gs, ss, bs


# (torch.autograd.grad: the gradients are returned, not accumulated into .grad -- what a step after
# optimizer.zero_grad(set_to_none=True) does; accumulating would add one framework `add` kernel per tensor to both columns)
def step_single(xs, gs, ss, bs):
    ys = [lsq(x, s, b, axis=0, is_perchannel=True, **KW) for x, s, b in zip(xs, ss, bs)]
    return torch.autograd.grad(ys, xs + ss, gs)


def step_fused(xs, gs, ss, bs):
    ys = lsq_foreach(xs, ss, bs, axis=0, **KW)
    return torch.autograd.grad(ys, xs + ss, gs)


# the backend ops alone (what bench.py times for the single-tensor workloads): no autograd graph, forward op + backward op
TAIL = (-128, 127, -128, 127, True, 1.0, True, False, False)


def ops_single(xs, gs, ss, bs):
    ops = torch.ops.torchlsq_native if E.host_binding() == "native" else torch.ops.torchlsq
    out = []
    for x, g, s, b in zip(xs, gs, ss, bs):
        y = ops.lsq_forward_per_channel(x, s, b, 0, *TAIL)
        out.append(ops.lsq_backward_per_channel(g, x, s, b, 0, *TAIL))
    return out


def ops_fused(xs, gs, ss, bs):
    n = len(xs)
    if E.host_binding() == "native":
        nat = torch.ops.torchlsq_native
        ys = nat.lsq_forward_per_channel_multi(xs, ss, bs, [0] * n, *TAIL)
        return nat.lsq_backward_per_channel_multi(gs, xs, ss, bs, [0] * n, *TAIL)
    ys = E.hip_forward_per_channel_multi(xs, ss, bs, [0] * n, *TAIL)
    return E.hip_backward_per_channel_multi(gs, xs, ss, bs, [0] * n, *TAIL)


def wall(fn, args, reps=30):
    for _ in range(5):
        fn(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(*args)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def gpu(fn, args, reps=10):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn(*args)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps):
                fn(*args)
        gr.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    print("# tools/exp_foreach.py on one MI355X: N per-channel qint8 weight quantizers (axis 0, symmetric), forward + backward per step")
    print("# single = N x (lsq forward + backward), fused = one lsq_foreach (one launch per 32 tensors each way); us per step")
    print("# autograd = through functional.lsq / lsq_foreach and torch.autograd.grad; ops only = the backend ops called directly (as bench.py does)")
    resnetish = [(64, 64, 3, 3)] * 4 + [(128, 128, 3, 3)] * 4 + [(256, 256, 3, 3)] * 6 + [(512, 512, 3, 3)] * 3 + \
                [(128, 64, 3, 3), (256, 128, 3, 3), (512, 256, 3, 3), (1000, 512)]
    vit_block = [(2304, 768), (768, 768), (3072, 768), (768, 3072)] * 12
    cases = [("50 x [512,512,3,3] fp32 (BASELINE config 3 x 50)", [(512, 512, 3, 3)] * 50, torch.float32),
             ("50 x [512,512,3,3] bf16", [(512, 512, 3, 3)] * 50, torch.bfloat16),
             ("ResNet-18-like conv stack, 21 weights fp32", resnetish, torch.float32),
             ("ViT-B linear weights, 48 tensors fp32", vit_block, torch.float32)]
    have_native = E.native_lsq() is not None
    for name, shapes, dtype in cases:
        for binding in ("native", "ctypes"):
            if binding == "native" and not have_native:
                continue
            E.set_host_binding(binding)
            args = make(shapes, dtype)
            fusable = sum(E.hip_multi_eligible(x, 0) for x in args[0])
            elems = sum(x.numel() for x in args[0])
            ws, wf = wall(step_single, args), wall(step_fused, args)
            gs_, gf = gpu(step_single, args), gpu(step_fused, args)
            esz = args[0][0].element_size()
            line = "%-50s %-7s fusable %2d/%2d | autograd, wall: single %8.1f  fused %8.1f  (%.2fx) | GPU: single %8.1f  fused %8.1f  (%.2fx, %.2f TB/s)" % (
                name, binding, fusable, len(shapes), ws, wf, ws / wf, gs_, gf, gs_ / gf, 5 * esz * elems / gf / 1e6)
            if fusable == len(shapes):
                with torch.no_grad():
                    plain = [[t.detach() for t in lst] for lst in args]
                    os_, of = wall(ops_single, plain), wall(ops_fused, plain)
                line += " | ops only, wall: single %8.1f  fused %8.1f  (%.2fx)" % (os_, of, os_ / of)
            print(line, flush=True)
            del args
        E.set_host_binding("native" if have_native else "ctypes")


if __name__ == "__main__":
    main()
