"""A timed parity SOAK of the shipped library (liblsq_hip.so + the C++ torch binding, no tools build, no knobs) against the CPU
oracle at the sizes the launch policy was written for.

    python3 tools/soak_parity.py --minutes 15 [--seed 1] > profiles/rNN_soak.txt

tests/test_fuzz_gpu.py draws shapes of at most 600 000 elements (the oracle has to keep a test fast), where only the
256-lane windows and the segment walk are the policy's own choice and the other families are reached by forcing them.  Here
the draws are 0.3-24 M elements in the layouts a network produces -- NCHW / NHWC activations, token matrices, conv / linear
weights, [B, C, L] sequences -- with awkward extents on purpose (channel counts that are no multiple of a packet, odd inner
sizes, prime row counts, one-element-offset storage), all four storage types, all mode combinations; every case runs the
forward and the backward through `torchlsq.functional.lsq` as a model does and is held to the bars of the parity tests (y, dx
bit-exact; d_scale / d_shift within 1e-6 of sum|terms|).  The report counts the cases per kernel family as the plan query
(lsq_hip_plan_backward_per_channel) names it, so that "0 mismatches" says which code it covers.  A mismatch is printed with
everything needed to replay it (seed, case number) and the run goes on; exit code 1 if there was any.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))

from helpers import assert_bits_equal, assert_reduction_close  # noqa: E402
from oracle import lsq_oracle as O  # noqa: E402

RANGES = [(0, 127, 0, 255), (-64, 63, -128, 127), (-128, 127, -128, 127), (0, 255, 0, 255), (-8, 7, -128, 127), (0, 15, 0, 255)]
ODD = [3, 5, 7, 9, 11, 13, 14, 17, 19, 23, 28, 31, 49, 56]
CH = [3, 24, 48, 64, 96, 100, 128, 160, 192, 197, 256, 320, 384, 512, 640, 768, 1000, 1024, 1280, 2048, 3072, 4096]
ROWS = [197, 577, 1009, 1024, 3152, 4096, 6151, 8192, 12608, 16384, 16400, 25216, 32768]


def draw_shape(rng):
    """(shape, axis, label); 0.3-24 M elements"""
    for _ in range(1000):
        kind = int(rng.integers(0, 7))
        if kind == 0:       # NCHW activation
            hw = int(rng.choice(ODD))
            s, ax, lab = (int(rng.choice([1, 2, 8, 16, 32, 33, 64, 128, 256])), int(rng.choice(CH)), hw, hw), 1, "nchw"
        elif kind == 1:     # tokens x features, quantized along the features
            s, ax, lab = (int(rng.choice(ROWS)), int(rng.choice(CH))), 1, "tokens"
        elif kind == 2:     # [B, T, F]
            s, ax, lab = (int(rng.choice([8, 16, 64])), int(rng.choice([49, 197, 256, 577])), int(rng.choice(CH))), 2, "btf"
        elif kind == 3:     # conv weight, per output channel
            k = int(rng.choice([1, 3, 5, 7]))
            s, ax, lab = (int(rng.choice(CH)), int(rng.choice(CH)), k, k), 0, "conv-w"
        elif kind == 4:     # linear weight, per output row
            s, ax, lab = (int(rng.choice(CH)), int(rng.choice([576, 768, 1000, 2304, 3072, 4096, 9216]))), 0, "linear-w"
        elif kind == 5:     # NHWC
            hw = int(rng.choice(ODD))
            s, ax, lab = (int(rng.choice([1, 8, 16, 32])), hw, hw, int(rng.choice(CH))), 3, "nhwc"
        else:               # [B, C, L]
            s, ax, lab = (int(rng.choice([4, 16, 64])), int(rng.choice(CH)), int(rng.choice([100, 1000, 1023, 4096, 16000]))), 1, "bcl"
        n = int(np.prod(s))
        if 300_000 <= n <= 24_000_000:
            return s, ax, lab
    raise RuntimeError("no shape drawn")


def narrow_equal(got, want, x, what):
    """16-bit storage: bit patterns, a mismatch names its first element.  A NaN matches any NaN: the payload is not part of
    the bar (torch's own fp32 -> bf16 conversion on the CPU writes 0xffff or 0x7fc0 depending on whether the element went
    through its vector or its scalar loop), which is where the EXPECTED bits of a NaN come from here.  That init_mode hands
    the storage bits of x through, NaN payloads included, is tests/test_init_passthrough_gpu.py's business."""
    a, b = got.contiguous().view(torch.int16).reshape(-1), want.contiguous().view(torch.int16).reshape(-1)
    if not torch.equal(a, b):
        both_nan = torch.isnan(got.reshape(-1).float()) & torch.isnan(want.reshape(-1).float())
        bad = torch.nonzero((a != b) & ~both_nan).reshape(-1)
        if bad.numel() == 0:
            return
        i = int(bad[0])
        raise AssertionError("%s: %d of %d elements differ; first at %d: got 0x%04x want 0x%04x (x there %r)" % (
            what, bad.numel(), a.numel(), i, int(a[i]) & 0xffff, int(b[i]) & 0xffff, float(x.reshape(-1)[i])))


EXTREME = [0.0]
AGAINST = ["oracle"]
REF = [None]          # --against reference: the worker process that runs the reference's own CPU ops


def ref_worker():
    """`--ref-worker` (started by --against reference): the REFERENCE's CPU ops -- oracle/_ref/libtorchlsq_ref_ops.so, its four
    CPU translation units built where they lie by oracle/build_ref.py and shipped with the snapshot -- in a process of their
    own (the library registers the same torchlsq::* names as the product).  One request per line on stdin: a directory with
    x / g / scale / shift as .npy and the arguments as JSON; y / dx / d_scale / d_shift go back into it."""
    import json
    so = os.path.join(ROOT, "oracle", "_ref", "libtorchlsq_ref_ops.so")
    torch.ops.load_library(so)
    ref = torch.ops.torchlsq
    print("ready", flush=True)
    for line in sys.stdin:
        d = line.strip()
        if not d:
            break
        try:
            a = json.load(open(os.path.join(d, "args.json")))
            x = torch.from_numpy(np.load(os.path.join(d, "x.npy"))).requires_grad_(True)
            g = torch.from_numpy(np.load(os.path.join(d, "g.npy")))
            sc = torch.from_numpy(np.load(os.path.join(d, "scale.npy"))).requires_grad_(True)
            sh = torch.from_numpy(np.load(os.path.join(d, "shift.npy"))).requires_grad_(True)
            y = ref.lsq(x, sc, sh, a["qmin"], a["qmax"], a["tmin"], a["tmax"], a["axis"], a["use_gs"], a["gs"], a["affine"],
                        a["per_channel"], a["eval_mode"], a["init_mode"])
            y.backward(g)
            np.save(os.path.join(d, "y.npy"), y.detach().numpy())
            np.save(os.path.join(d, "dx.npy"), x.grad.numpy())
            np.save(os.path.join(d, "ds.npy"), sc.grad.numpy() if sc.grad is not None else np.zeros(0))
            np.save(os.path.join(d, "db.npy"), sh.grad.numpy() if sh.grad is not None else np.zeros(0))
            print("ok", flush=True)
        except Exception as e:          # the reference's own TORCH_CHECKs included
            print("error " + repr(e).replace("\n", " ")[:300], flush=True)


def ref_call(xs, gsh, scale, shift, args):
    """(y, dx, ds, db) of the reference's CPU ops for these inputs, through the worker"""
    import json
    import shutil
    import subprocess
    import tempfile
    if REF[0] is None:
        env = {k: v for k, v in os.environ.items() if not k.startswith("HIP_") and k != "ROCR_VISIBLE_DEVICES"}
        REF[0] = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--ref-worker"], stdin=subprocess.PIPE,
                                  stdout=subprocess.PIPE, text=True, env=env)
        assert REF[0].stdout.readline().strip() == "ready", "the reference worker did not start (oracle/_ref missing?)"
    d = tempfile.mkdtemp(prefix="lsqsoak", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for name, arr in (("x", xs), ("g", gsh), ("scale", scale), ("shift", shift)):
            np.save(os.path.join(d, name + ".npy"), np.ascontiguousarray(arr))
        json.dump(args, open(os.path.join(d, "args.json"), "w"))
        REF[0].stdin.write(d + "\n")
        REF[0].stdin.flush()
        reply = REF[0].stdout.readline().strip()
        assert reply == "ok", "reference worker: " + reply
        return tuple(np.load(os.path.join(d, n + ".npy")) for n in ("y", "dx", "ds", "db"))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def one_case(rng, dev, lsq, E, counts):
    extreme = rng.random() < EXTREME[0]
    against_ref = AGAINST[0] == "reference"
    shape, axis, lab = draw_shape(rng)
    n = int(np.prod(shape))
    dtype = [torch.float32, torch.float32, torch.bfloat16, torch.bfloat16, torch.float16, torch.float64][int(rng.integers(0, 6))]
    if dtype == torch.float64 and n > 6_000_000:
        dtype = torch.float32
    if against_ref and dtype in (torch.bfloat16, torch.float16):      # the reference's CPU ops: AT_DISPATCH_FLOATING_TYPES
        dtype = torch.float32
    npdt = np.float64 if dtype == torch.float64 else np.float32
    narrow = dtype in (torch.bfloat16, torch.float16)
    per_channel = rng.random() < 0.85
    C = shape[axis] if per_channel else 1
    qmin, qmax, tmin, tmax = RANGES[int(rng.integers(0, len(RANGES)))]
    affine = bool(rng.random() < 0.6) or not (qmin <= 0 <= qmax)
    eval_mode = bool(rng.random() < 0.1)
    init_mode = bool(rng.random() < 0.1)
    use_gs = bool(rng.random() < 0.8)
    gs = float(rng.choice([1.0, 0.5, 3.0]))
    step = float(rng.choice([0.003, 0.05, 0.4]))
    offset = bool(rng.random() < 0.12)
    tag = "%s %s %s pc=%s axis=%d q=(%d,%d,%d,%d) affine=%s eval=%s init=%s gs=(%s,%s) offset=%s" % (
        lab, shape, str(dtype).replace("torch.", ""), per_channel, axis, qmin, qmax, tmin, tmax, affine, eval_mode, init_mode, use_gs, gs, offset)

    x = rng.standard_normal(n, dtype=np.float32).astype(npdt) * npdt(step * (qmax - qmin) * 0.4) + npdt(step * (qmax + qmin) * 0.5)
    k = min(n, 64)
    x[rng.integers(0, n, size=k)] = rng.choice(np.array([0.0, step * qmin, step * qmax, step * (qmin - 0.5), step * (qmax + 0.5),
                                                         step * 0.5, -step * 0.5, np.inf, -np.inf, np.nan], dtype=npdt), size=k)
    g = rng.standard_normal(n, dtype=np.float32).astype(npdt) * npdt(1e-2)
    if narrow:
        x = torch.from_numpy(x).to(dtype).to(torch.float32).numpy()
        g = torch.from_numpy(g).to(dtype).to(torch.float32).numpy()
    scale = (rng.uniform(0.5, 1.5, size=C) * step).astype(npdt)
    if rng.random() < 0.3:
        scale[int(rng.integers(0, C))] *= -1.0
    if rng.random() < 0.1:
        scale[int(rng.integers(0, C))] = 0.0
    shift = (rng.standard_normal(C) * step * (2.0 if affine else 0.0)).astype(npdt)
    if extreme:         # --extreme: parameters and gradients no training run should see, a share of the cases
        what = int(rng.integers(0, 3))
        with np.errstate(over="ignore"):
            if what == 0:
                scale[int(rng.integers(0, C))] = npdt(rng.choice([1e-40, 1e30, np.inf, np.nan, 1e-30]))
            elif what == 1 and affine:
                shift[int(rng.integers(0, C))] = npdt(rng.choice([np.inf, -np.inf, np.nan, 1e30, -1e30]))
            else:
                g[rng.integers(0, n, size=8)] = rng.choice(np.array([np.inf, -np.inf, np.nan, 1e30, -1e30], dtype=npdt), size=8)
                if narrow:
                    g = torch.from_numpy(g).to(dtype).to(torch.float32).numpy()
        tag += " extreme=%d" % what
    xs, gsh = x.reshape(shape), g.reshape(shape)

    def on_gpu(a):
        t = torch.from_numpy(a).to(dev).to(dtype)
        if offset:          # the same values one element into their storage: 2/4/8-byte aligned only
            flat = torch.empty(t.numel() + 1, dtype=dtype, device=dev)[1:]
            flat.copy_(t.reshape(-1))
            t = flat.view(t.shape)
        return t

    xt = on_gpu(xs).requires_grad_(True)
    gt = on_gpu(gsh)
    st = torch.from_numpy(scale).to(dev).requires_grad_(True)
    bt = torch.from_numpy(shift).to(dev).requires_grad_(True)
    family = "per-tensor"
    if per_channel:
        plan = E.hip_plan_backward_per_channel(xt.detach(), axis, not affine, eval_mode, init_mode)
        family = "%s/%d" % (plan["kind"], plan["block"]) + ("/ring" if plan["ring_depth"] else "")
    counts[family] = counts.get(family, 0) + 1
    y = lsq(xt, st, bt, qmin, qmax, tmin, tmax, axis, use_gs, gs, affine, per_channel, eval_mode, init_mode)
    y.backward(gt)
    torch.cuda.synchronize()

    if per_channel:
        outer, C_, inner = O.axis_to_ocl(shape, axis)
        oy = O.fwd_pc(xs, scale, shift, outer, C_, inner, qmin, qmax, tmin, tmax, init_mode)
        r = O.bwd_pc(gsh, xs, scale, shift, outer, C_, inner, qmin, qmax, tmin, tmax, use_gs, gs, not affine, eval_mode, init_mode)
    else:
        oy = O.fwd_pt(xs, scale[0], shift[0], qmin, qmax, tmin, tmax, init_mode)
        r = O.bwd_pt(gsh, xs, scale[0], shift[0], qmin, qmax, tmin, tmax, use_gs, gs, not affine, eval_mode, init_mode)
    tag = "[%s] %s" % (family, tag)
    if against_ref:
        # the same inputs through the reference's own CPU ops: y, dx bit for bit; d_scale / d_shift inside the bar, with the
        # oracle's sum|terms| as the yardstick (the reference does not report one)
        ry, rdx, rds, rdb = ref_call(xs, gsh, scale, shift, dict(qmin=qmin, qmax=qmax, tmin=tmin, tmax=tmax, axis=axis, use_gs=use_gs, gs=gs,
                                                                 affine=affine, per_channel=per_channel, eval_mode=eval_mode, init_mode=init_mode))
        assert_bits_equal(y.detach().cpu().numpy(), ry, tag + " y vs the reference")
        assert_bits_equal(xt.grad.cpu().numpy(), rdx, tag + " dx vs the reference")
        assert_bits_equal(oy, ry, tag + " ORACLE y vs the reference")
        assert_bits_equal(r.dx, rdx, tag + " ORACLE dx vs the reference")
        ds = st.grad.cpu().numpy() if st.grad is not None else np.zeros(C, npdt)
        db = bt.grad.cpu().numpy() if bt.grad is not None else np.zeros(C, npdt)
        if rds.size == C:
            assert_reduction_close(ds, rds, r.abs_ds, tag + " ds vs the reference")
        if rdb.size == C:
            assert_reduction_close(db, rdb, r.abs_db, tag + " db vs the reference")
        return n, tag
    if narrow:
        want_y = torch.from_numpy(np.ascontiguousarray(oy)).to(dtype)
        want_dx = torch.from_numpy(np.ascontiguousarray(r.dx)).to(dtype)
        narrow_equal(y.detach().cpu(), want_y, xs, tag + " y")
        narrow_equal(xt.grad.cpu(), want_dx, xs, tag + " dx")
    else:
        assert_bits_equal(y.detach().cpu().numpy(), oy, tag + " y")
        assert_bits_equal(xt.grad.cpu().numpy(), r.dx, tag + " dx")
    ds = st.grad.cpu().numpy() if st.grad is not None else np.zeros(C, npdt)
    db = bt.grad.cpu().numpy() if bt.grad is not None else np.zeros(C, npdt)
    assert_reduction_close(ds, r.ds_wide, r.abs_ds, tag + " ds")
    assert_reduction_close(db, r.db_wide, r.abs_db, tag + " db")
    return n, tag


def side_case(rng, dev, lsq, E, counts):
    """the ops beside the reference's four (tests/test_fuzz_gpu.py::test_random_cases_of_the_side_outputs at these sizes):
    integer levels, the eval-mode mask + backward_from_mask, one-pass min / max (exact) and mean / std (1e-6)"""
    ops = torch.ops.torchlsq
    shape, axis, lab = draw_shape(rng)
    n = int(np.prod(shape))
    dtype = torch.float64 if (rng.random() < 0.25 and n <= 6_000_000) else torch.float32
    npdt = np.float64 if dtype == torch.float64 else np.float32
    per_channel = rng.random() < 0.8
    C = shape[axis] if per_channel else 1
    qmin, qmax, tmin, tmax = RANGES[int(rng.integers(0, len(RANGES)))]
    step = float(rng.choice([0.003, 0.05, 0.4]))
    offset = bool(rng.random() < 0.12)
    tag = "[side] %s %s %s pc=%s axis=%d q=(%d,%d,%d,%d) offset=%s" % (lab, shape, str(dtype).replace("torch.", ""), per_channel, axis,
                                                                      qmin, qmax, tmin, tmax, offset)
    x = (rng.standard_normal(n, dtype=np.float32).astype(npdt) * npdt(step * (qmax - qmin) * 0.4) + npdt(step * (qmax + qmin) * 0.5)).reshape(shape)
    g = (rng.standard_normal(n, dtype=np.float32).astype(npdt) * npdt(1e-2)).reshape(shape)
    scale = (rng.uniform(0.5, 1.5, size=C) * step).astype(npdt)
    shift = (rng.standard_normal(C) * step * 2.0).astype(npdt)

    def on_gpu(a):
        t = torch.from_numpy(a).to(dev)
        if offset:
            flat = torch.empty(t.numel() + 1, dtype=dtype, device=dev)[1:]
            flat.copy_(t.reshape(-1))
            t = flat.view(t.shape)
        return t

    xt, gt = on_gpu(x), on_gpu(g)
    st, bt = torch.from_numpy(scale).to(dev), torch.from_numpy(shift).to(dev)
    bias = 128 if qmax > 127 else 0
    counts["side/" + ("per-channel" if per_channel else "per-tensor")] = counts.get("side/" + ("per-channel" if per_channel else "per-tensor"), 0) + 1
    if per_channel:
        outer, C_, inner = O.axis_to_ocl(shape, axis)
        want_q = O.levels_pc(x, scale, shift, outer, C_, inner, qmin, qmax, tmin, tmax)
        want_y = O.fwd_pc(x, scale, shift, outer, C_, inner, qmin, qmax, tmin, tmax)
        r = O.bwd_pc(g, x, scale, shift, outer, C_, inner, qmin, qmax, tmin, tmax, True, 1.0, False, True, False)
        y, q = ops.lsq_quantize_per_channel(xt, st, bt, axis, qmin, qmax, tmin, tmax, bias)
        y2, mask = E.hip_forward_per_channel(xt, st, bt, axis, qmin, qmax, tmin, tmax, True, 1.0, False, True, False, want_mask=True)
        mn, mx = ops.lsq_minmax_per_channel(xt, axis)
        mu, sd = ops.lsq_meanstd_per_channel(xt, axis)
        moved = np.moveaxis(x, axis, 0).reshape(C, -1)
        want_mu, want_sd = O.meanstd(x, outer, C_, inner)
    else:
        want_q = O.levels_pt(x, scale[0], shift[0], qmin, qmax, tmin, tmax)
        want_y = O.fwd_pt(x, scale[0], shift[0], qmin, qmax, tmin, tmax)
        r = O.bwd_pt(g, x, scale[0], shift[0], qmin, qmax, tmin, tmax, True, 1.0, False, True, False)
        y, q = ops.lsq_quantize_per_tensor(xt, st, bt, qmin, qmax, tmin, tmax, bias)
        y2, mask = E.hip_forward_per_tensor(xt, st, bt, qmin, qmax, tmin, tmax, True, 1.0, False, True, False, want_mask=True)
        mn, mx = ops.lsq_minmax_per_tensor(xt)
        mu, sd = ops.lsq_meanstd_per_tensor(xt)
        moved = x.reshape(1, -1)
        want_mu, want_sd = O.meanstd(x, 1, 1, n)
    assert q.dtype == torch.int8 and q.shape == xt.shape, tag + " levels type / shape"
    assert np.array_equal(q.cpu().numpy().astype(np.int32) + bias, want_q.reshape(shape)), tag + " levels"
    assert_bits_equal(y.cpu().numpy(), want_y, tag + " y (quantize op)")
    assert_bits_equal(y2.cpu().numpy(), want_y, tag + " y (masked forward)")
    dx = ops.lsq_backward_from_mask(gt, mask)
    assert_bits_equal(dx.cpu().numpy(), r.dx, tag + " dx from mask")
    assert np.array_equal(mn.cpu().numpy().reshape(-1), moved.min(axis=1)), tag + " min"
    assert np.array_equal(mx.cpu().numpy().reshape(-1), moved.max(axis=1)), tag + " max"
    rtol = 1e-12 if dtype == torch.float64 else 1e-6
    spread = float(np.abs(moved).max()) + 1e-30
    try:
        np.testing.assert_allclose(mu.cpu().numpy().reshape(-1), want_mu, rtol=rtol, atol=rtol * spread, err_msg=tag + " mean")
        np.testing.assert_allclose(sd.cpu().numpy().reshape(-1), want_sd, rtol=rtol, atol=0, equal_nan=True, err_msg=tag + " std")
    except AssertionError as e:
        raise AssertionError(tag + " moments: " + str(e).replace("\n", " ")[:300])
    return n, tag


def foreach_case(rng, dev, lsq, E, counts):
    """`lsq_foreach` over a random list of weight-like tensors (fusable and not, mixed shapes, one or two storage types)
    against the same tensors through N single `lsq` calls: outputs and all three gradients bit for bit, as its docstring
    says; the single calls are what --ops lsq holds to the oracle"""
    from torchlsq.functional import lsq_foreach
    n_t = int(rng.choice([2, 3, 5, 8, 31, 32, 33, 50, 70]))
    dtypes = [[torch.float32], [torch.bfloat16], [torch.float16], [torch.float32, torch.bfloat16]][int(rng.integers(0, 4))]
    qmin, qmax, tmin, tmax = RANGES[int(rng.integers(0, len(RANGES)))]
    affine = bool(rng.random() < 0.5) or not (qmin <= 0 <= qmax)
    eval_mode = bool(rng.random() < 0.1)
    init_mode = bool(rng.random() < 0.1)
    use_gs = bool(rng.random() < 0.8)
    gs = float(rng.choice([1.0, 0.5, 3.0]))
    step = 0.05
    xs, gts, scs, shs, axes, total = [], [], [], [], [], 0
    for _ in range(n_t):
        kind = int(rng.integers(0, 5))
        if kind == 0:
            k = int(rng.choice([1, 3, 5]))
            shape, ax = (int(rng.choice([16, 24, 64, 100, 128, 256, 512])), int(rng.choice([3, 16, 64, 128, 256])), k, k), 0
        elif kind == 1:
            shape, ax = (int(rng.choice([10, 64, 100, 256, 768, 1000])), int(rng.choice([64, 100, 257, 576, 768, 1024, 3072]))), 0
        elif kind == 2:     # quantized along the input features
            shape, ax = (int(rng.choice([64, 256, 768])), int(rng.choice([64, 256, 768, 1000]))), 1
        elif kind == 3:     # tiny
            shape, ax = (int(rng.choice([1, 2, 8])), int(rng.choice([1, 3, 9, 27]))), 0
        else:               # depthwise-like: short channel rows
            shape, ax = (int(rng.choice([32, 96, 384])), 1, 3, 3), 0
        dt = dtypes[int(rng.integers(0, len(dtypes)))]
        n = int(np.prod(shape))
        total += n
        x = torch.from_numpy((rng.standard_normal(n, dtype=np.float32) * np.float32(step * (qmax - qmin) * 0.4)
                              + np.float32(step * (qmax + qmin) * 0.5)).reshape(shape)).to(dev).to(dt)
        g = torch.from_numpy((rng.standard_normal(n, dtype=np.float32) * np.float32(1e-2)).reshape(shape)).to(dev).to(dt)
        C = shape[ax]
        xs.append(x); gts.append(g); axes.append(ax)
        scs.append(torch.from_numpy((rng.uniform(0.5, 1.5, size=C) * step).astype(np.float32)).to(dev))
        shs.append(torch.from_numpy((rng.standard_normal(C) * step * (2.0 if affine else 0.0)).astype(np.float32)).to(dev))
    unused = set(int(i) for i in rng.integers(0, n_t, size=int(rng.integers(0, 3))))     # outputs nobody uses: no gradient at all
    tag = "[foreach] %d tensors %s q=(%d,%d,%d,%d) affine=%s eval=%s init=%s gs=(%s,%s) unused=%s" % (
        n_t, "+".join(str(d).replace("torch.", "") for d in dtypes), qmin, qmax, tmin, tmax, affine, eval_mode, init_mode, use_gs, gs, sorted(unused))
    counts["foreach/%s" % ("+".join(str(d).replace("torch.", "") for d in dtypes))] = counts.get(
        "foreach/%s" % ("+".join(str(d).replace("torch.", "") for d in dtypes)), 0) + 1

    def leaves(ts):
        return [t.detach().clone().requires_grad_(True) for t in ts]

    fused = sum(1 for i in range(n_t) if E.hip_multi_eligible(xs[i], axes[i]))
    counts["  tensors the multi-tensor kernels take"] = counts.get("  tensors the multi-tensor kernels take", 0) + fused
    counts["  tensors that go through single calls"] = counts.get("  tensors that go through single calls", 0) + n_t - fused
    xa, sa, ba = leaves(xs), leaves(scs), leaves(shs)
    ys = lsq_foreach(xa, sa, ba, qmin, qmax, tmin, tmax, axes, use_gs, gs, affine, eval_mode, init_mode)
    live = [i for i in range(n_t) if i not in unused]
    torch.autograd.backward([ys[i] for i in live], [gts[i] for i in live])
    xb, sb, bb = leaves(xs), leaves(scs), leaves(shs)
    for i in range(n_t):
        y1 = lsq(xb[i], sb[i], bb[i], qmin, qmax, tmin, tmax, axes[i], use_gs, gs, affine, True, eval_mode, init_mode)
        if i in live:
            y1.backward(gts[i])
        what = tag + " tensor %d %s axis %d" % (i, tuple(xs[i].shape), axes[i])
        assert torch.equal(ys[i].detach().view(torch.uint8), y1.detach().view(torch.uint8)), what + " y"
        for name, a, b in (("dx", xa[i].grad, xb[i].grad), ("d_scale", sa[i].grad, sb[i].grad), ("d_shift", ba[i].grad, bb[i].grad)):
            assert (a is None) == (b is None), what + " %s: one route has a gradient, the other none" % name
            if a is not None:
                same = torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8)) or torch.equal(
                    torch.nan_to_num(a.float(), nan=12345.0), torch.nan_to_num(b.float(), nan=12345.0))
                assert same, what + " " + name
    torch.cuda.synchronize()
    return total, tag


def shards_case(rng, dev, lsq, E, counts):
    """The batch-sharded backward without a transport: dim 0 cut into 2-8 UNEVEN contiguous shards, every shard through
    `sharded_backward(..., global_numel=<the whole tensor's>, reduce=False)` -- the rank-local call of
    torchlsq.distributed, wide fp64 sums scaled with the global count -- the shards' sums added in fp64 and rounded once, as
    the all-reduce + cast does; against the oracle on the WHOLE tensor: dx bit-exact, d_scale / d_shift inside the bar."""
    from torchlsq import distributed as D
    for _ in range(100):
        shape, axis, lab = draw_shape(rng)
        if axis != 0 and shape[0] >= 8:
            break
    n = int(np.prod(shape))
    dtype = [torch.float32, torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 4))]
    narrow = dtype != torch.float32
    per_channel = rng.random() < 0.8
    C = shape[axis] if per_channel else 1
    qmin, qmax, tmin, tmax = RANGES[int(rng.integers(0, len(RANGES)))]
    affine = bool(rng.random() < 0.6) or not (qmin <= 0 <= qmax)
    init_mode = bool(rng.random() < 0.1)
    use_gs = bool(rng.random() < 0.85)
    gs = float(rng.choice([1.0, 0.5, 3.0]))
    step = float(rng.choice([0.003, 0.05, 0.4]))
    k = int(rng.integers(2, 9))
    cuts = sorted(set(int(c) for c in rng.integers(0, shape[0] + 1, size=k - 1)))        # empty shards allowed
    bounds = [0] + cuts + [shape[0]]
    tag = "[shards] %s %s %s pc=%s axis=%d q=(%d,%d,%d,%d) affine=%s init=%s gs=(%s,%s) rows per shard %s" % (
        lab, shape, str(dtype).replace("torch.", ""), per_channel, axis, qmin, qmax, tmin, tmax, affine, init_mode, use_gs, gs,
        [b - a for a, b in zip(bounds[:-1], bounds[1:])])
    x = rng.standard_normal(n, dtype=np.float32) * np.float32(step * (qmax - qmin) * 0.4) + np.float32(step * (qmax + qmin) * 0.5)
    g = rng.standard_normal(n, dtype=np.float32) * np.float32(1e-2)
    if narrow:
        x = torch.from_numpy(x).to(dtype).to(torch.float32).numpy()
        g = torch.from_numpy(g).to(dtype).to(torch.float32).numpy()
    xs, gsh = x.reshape(shape), g.reshape(shape)
    scale = (rng.uniform(0.5, 1.5, size=C) * step).astype(np.float32)
    shift = (rng.standard_normal(C) * step * (2.0 if affine else 0.0)).astype(np.float32)
    xt, gt = torch.from_numpy(xs).to(dev).to(dtype), torch.from_numpy(gsh).to(dev).to(dtype)
    st, bt = torch.from_numpy(scale).to(dev), torch.from_numpy(shift).to(dev)
    counts["shards/%d" % (len(bounds) - 1)] = counts.get("shards/%d" % (len(bounds) - 1), 0) + 1
    dxs, total = [], None
    for a, b in zip(bounds[:-1], bounds[1:]):
        dx, ds_, db_ = D.sharded_backward(gt[a:b], xt[a:b], st, bt, qmin, qmax, tmin, tmax, axis, use_gs, gs, affine, per_channel,
                                         False, init_mode, None, n, False, False)
        dxs.append(dx)
        # reduce=False hands back the rank-local sums ALREADY rounded to the parameter type; the transport adds fp64 sums, so
        # take those: the same call with the wide output
        ops = torch.ops.torchlsq_native
        if b > a:
            if per_channel:
                _dx, wide = ops.lsq_backward_per_channel_wide(gt[a:b], xt[a:b], st, bt, axis, qmin, qmax, tmin, tmax, use_gs, gs,
                                                              not affine, False, init_mode, n)
            else:
                _dx, wide = ops.lsq_backward_per_tensor_wide(gt[a:b], xt[a:b], st, bt, qmin, qmax, tmin, tmax, use_gs, gs,
                                                             not affine, False, init_mode, n)
            assert torch.equal(_dx.view(torch.uint8), dx.view(torch.uint8)), tag + " dx of the two calls"
            total = wide.double().reshape(2, -1) if total is None else total + wide.double().reshape(2, -1)
    ds = total[0].to(torch.float32).cpu().numpy()
    db = total[1].to(torch.float32).cpu().numpy()
    dx_all = torch.cat(dxs, dim=0)
    torch.cuda.synchronize()
    if per_channel:
        outer, C_, inner = O.axis_to_ocl(shape, axis)
        r = O.bwd_pc(gsh, xs, scale, shift, outer, C_, inner, qmin, qmax, tmin, tmax, use_gs, gs, not affine, False, init_mode)
    else:
        r = O.bwd_pt(gsh, xs, scale[0], shift[0], qmin, qmax, tmin, tmax, use_gs, gs, not affine, False, init_mode)
    if narrow:
        narrow_equal(dx_all.cpu(), torch.from_numpy(np.ascontiguousarray(r.dx)).to(dtype), xs, tag + " dx")
    else:
        assert_bits_equal(dx_all.cpu().numpy(), r.dx, tag + " dx")
    assert_reduction_close(ds, r.ds_wide, r.abs_ds, tag + " ds")
    assert_reduction_close(db, r.db_wide, r.abs_db, tag + " db")
    return n, tag


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ops", choices=["lsq", "side", "foreach", "shards"], default="lsq", help="lsq: forward + backward through functional.lsq; "
                    "side: levels, mask backward, min / max, mean / std; foreach: lsq_foreach against N single calls; "
                    "shards: dim 0 cut into uneven shards, their fp64 sums added, against the oracle on the whole tensor")
    ap.add_argument("--only", type=int, default=-1, help="replay: run this case number of the seed only")
    ap.add_argument("--against", choices=["oracle", "reference"], default="oracle", help="--ops lsq: reference = the reference's own CPU "
                    "ops (oracle/_ref/libtorchlsq_ref_ops.so, in a worker process; fp32 / fp64 storage only)")
    ap.add_argument("--ref-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--extreme", type=float, default=0.0, help="--ops lsq: share of the cases with a subnormal / huge / inf / NaN scale, "
                    "an inf / NaN shift or inf / NaN / 1e30 gradients")
    a = ap.parse_args()
    if a.ref_worker:
        return ref_worker()
    AGAINST[0] = a.against
    import torchlsq  # noqa: F401
    from torchlsq import extension as E
    from torchlsq.functional import lsq
    E._assert_has_ops()
    assert E.host_binding() == "native", "the soak is of the shipped stack: C++ binding over liblsq_hip.so"
    dev = torch.device("cuda:0")
    EXTREME[0] = a.extreme
    counts, failures, elements, case = {}, [], 0, 0
    t_end = time.time() + a.minutes * 60.0
    while time.time() < t_end:
        rng = np.random.default_rng([a.seed, case])       # a case replays from (seed, case) alone
        if a.only >= 0 and case != a.only:
            case += 1
            continue
        try:
            n, _tag = {'lsq': one_case, 'side': side_case, 'foreach': foreach_case, 'shards': shards_case}[a.ops](rng, dev, lsq, E, counts)
            elements += n
        except AssertionError as e:
            failures.append("seed %d case %d: %s" % (a.seed, case, str(e)[:600]))
            print("MISMATCH " + failures[-1], flush=True)
        case += 1
        if a.only >= 0:
            break
    if REF[0] is not None:
        REF[0].stdin.close()
        REF[0].wait(timeout=60)
    print("# tools/soak_parity.py --minutes %g --seed %d --ops %s%s on %s: the shipped library %s %s"
          % (a.minutes, a.seed, a.ops, " --extreme %g" % a.extreme if a.extreme else "", torch.cuda.get_device_name(0),
             {"lsq": "through torchlsq.functional.lsq", "side": "(quantize ops, masked forward + backward_from_mask, observer statistics)",
              "foreach": "through torchlsq.functional.lsq_foreach",
              "shards": "through torchlsq.distributed.sharded_backward, shard by shard,"}[a.ops],
             "against N single calls" if a.ops == "foreach" else
             ("against the REFERENCE's CPU ops (oracle/_ref/libtorchlsq_ref_ops.so) -- and the oracle against them" if a.against == "reference"
              else "against oracle/lsq_oracle.c")))
    if a.ops == "foreach":
        print("# bar: every output and gradient bit-identical to N single lsq calls (which --ops lsq holds to the oracle)")
    elif a.ops == "shards":
        print("# bars: dx of the concatenated shards bit-exact; the shards' fp64 sums, added and rounded once, within 1e-6 of sum|terms|")
    elif a.ops == "lsq":
        print("# bars: y, dx bit-exact (16-bit storage: the fp32 result rounded to the storage type); d_scale, d_shift within 1e-6 of sum|terms|")
    else:
        print("# bars: int8 levels, y, dx-from-mask, min, max exact; mean / std within 1e-6 (fp32), 1e-12 (fp64)")
    print("cases %d   elements %.3g   mismatches %d" % (case if a.only < 0 else 1, elements, len(failures)))
    for fam in sorted(counts, key=lambda f: -counts[f]):
        print("  %-40s %7d%s" % (fam, counts[fam], "" if fam.startswith("  ") else " cases"))
    for f in failures:
        print("MISMATCH " + f)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
