#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REFERENCE's own CPU ops.

Run in the build container only (needs /root/reference):

    python oracle/build_ref.py            # reference csrc -> oracle/_ref/libtorchlsq_ref_ops.so
    python tests/golden/make_golden.py            # small cases + cfg1/cfg3/cfg5 digests (~2 min)
    python tests/golden/make_golden.py --big      # + cfg2 / cfg4 (205 M elements each, ~10 min)
    python tests/golden/make_golden.py --skip-small --shards   # only the per-rank records of the batch-sharded runs (~15 min)

What is written (data only -- inputs and the reference's outputs):
    tests/golden/small_cases.npz / small_cases.json   full tensors for ~60 small cases
    tests/golden/config_digests.json                  sha256 digests + ds/db for BASELINE configs

The expected outputs come from `torch.ops.torchlsq.*` of the reference library
(lsq.cpp:104-146 front op, lsq_autograd.cpp autograd, lsq_cpu.cpp CPU kernels).  This process
never imports the product package (it would register the same op names); synth.py is loaded by
path.  While generating, every case is also cross-checked against oracle/lsq_oracle.c so that a
drift between the restatement and the reference is caught at fixture time.
"""
import argparse
import hashlib
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import lsq_oracle as O  # noqa: E402
from oracle import build_ref  # noqa: E402


def _load_synth():
    p = os.path.join(ROOT, "lsqfakequantize-pytorch_amd", "torchlsq", "synth.py")
    spec = importlib.util.spec_from_file_location("_synth_by_path", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


S = _load_synth()


def load_reference():
    build_ref.build_all(verbose=False)
    torch.ops.load_library(build_ref.OPS_SO)
    return torch.ops.torchlsq


def ref_fwd_bwd(ref, x, g, scale, shift, p):
    """Run the reference front op + autograd.  Returns y, dx, ds, db as numpy."""
    x = x.clone().requires_grad_(True)
    scale = scale.clone().requires_grad_(True)
    shift = shift.clone().requires_grad_(True)
    y = ref.lsq(x, scale, shift, p["quant_min"], p["quant_max"], p["type_min"], p["type_max"], p["axis"],
                p["use_grad_scaling"], p["grad_scaler"], p["is_affine"], p["is_perchannel"],
                p["eval_mode"], p["init_mode"])
    y.backward(g)
    return (y.detach().numpy(), x.grad.numpy(), scale.grad.numpy(), shift.grad.numpy())


def oracle_fwd_bwd(x, g, scale, shift, p):
    xn, gn = x.numpy(), g.numpy()
    sym = not p["is_affine"]
    if p["is_perchannel"]:
        outer, C, inner = O.axis_to_ocl(xn.shape, p["axis"])
        sc = scale.numpy() if scale.numel() == C else np.repeat(scale.numpy(), C)
        sh = shift.numpy() if shift.numel() == C else np.repeat(shift.numpy(), C)
        y = O.fwd_pc(xn, sc, sh, outer, C, inner, p["quant_min"], p["quant_max"], p["type_min"],
                     p["type_max"], p["init_mode"])
        r = O.bwd_pc(gn, xn, sc, sh, outer, C, inner, p["quant_min"], p["quant_max"], p["type_min"],
                     p["type_max"], p["use_grad_scaling"], p["grad_scaler"], sym, p["eval_mode"],
                     p["init_mode"])
        ds, db, dsw, dbw, ads, adb = r.ds, r.db, r.ds_wide, r.db_wide, r.abs_ds, r.abs_db
        if scale.numel() != C:  # size-1 parameter repeated by the front op (lsq.cpp:124-126)
            ds, dsw, ads = (np.array([v.sum()], dtype=v.dtype) for v in (ds.astype(np.float64), dsw, ads))
            ds = ds.astype(xn.dtype)
        if shift.numel() != C:
            db, dbw, adb = (np.array([v.sum()], dtype=v.dtype) for v in (db.astype(np.float64), dbw, adb))
            db = db.astype(xn.dtype)
        return y, r.dx, ds, db, dsw, dbw, ads, adb
    y = O.fwd_pt(xn, scale[0].item(), shift[0].item(), p["quant_min"], p["quant_max"], p["type_min"],
                 p["type_max"], p["init_mode"])
    r = O.bwd_pt(gn, xn, scale[0].item(), shift[0].item(), p["quant_min"], p["quant_max"], p["type_min"],
                 p["type_max"], p["use_grad_scaling"], p["grad_scaler"], sym, p["eval_mode"], p["init_mode"])
    return y, r.dx, r.ds, r.db, r.ds_wide, r.db_wide, r.abs_ds, r.abs_db


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def check_against_oracle(name, ref_out, orc_out, tol=1e-6):
    y, dx, ds, db = ref_out
    oy, odx, ods, odb, dsw, dbw, ads, adb = orc_out
    assert bits_equal(y, oy.reshape(y.shape)), name + ": oracle y differs from the reference"
    assert bits_equal(dx, odx.reshape(dx.shape)), name + ": oracle dx differs from the reference"
    for r_, o_, a_ in ((ds, ods, ads), (db, odb, adb)):
        r64, o64 = r_.astype(np.float64), o_.astype(np.float64)
        scale_ = np.maximum(a_, 1e-300)
        bad = ~(np.abs(r64 - o64) <= tol * scale_) & ~(np.isnan(r64) & np.isnan(o64)) & ~(r64 == o64)
        assert not bad.any(), "%s: ds/db off: ref %r oracle %r sum|t| %r" % (name, r_[bad][:4], o_[bad][:4], a_[bad][:4])


# ---------------------------------------------------------------------------------------------
# small cases
# ---------------------------------------------------------------------------------------------
def P(**kw):
    p = dict(quant_min=0, quant_max=127, type_min=0, type_max=255, axis=1, use_grad_scaling=True,
             grad_scaler=1.0, is_affine=True, is_perchannel=False, eval_mode=False, init_mode=False)
    p.update(kw)
    return p


def small_case_specs():
    specs = []

    def add(name, shape, p, dtype="float32", scale=0.03, shift=0.0, x_mean=1.5, x_std=1.0, special=None,
            nscale=None, nshift=None):
        specs.append(dict(name=name, shape=list(shape), p=p, dtype=dtype, scale=scale, shift=shift,
                          x_mean=x_mean, x_std=x_std, special=special, nscale=nscale, nshift=nshift))

    act7 = dict()
    sym7 = dict(quant_min=-64, quant_max=63, type_min=-128, type_max=127, is_affine=False)
    sym8 = dict(quant_min=-128, quant_max=127, type_min=-128, type_max=127, is_affine=False)
    for dt in ("float32", "float64"):
        add("pt_affine7_" + dt, (2, 3, 5, 7), P(**act7), dt, shift=0.1)
        add("pt_affine8_" + dt, (2, 3, 5, 7), P(quant_max=255), dt, shift=-0.2)
        add("pt_sym7_" + dt, (4, 8, 6, 6), P(**sym7), dt, scale=0.02, x_mean=0.0)
        add("pt_eval_" + dt, (2, 3, 5, 7), P(eval_mode=True), dt, shift=0.1)
        add("pt_init_" + dt, (2, 3, 5, 7), P(init_mode=True), dt, shift=0.1)
        add("pt_noscaling_" + dt, (2, 3, 5, 7), P(use_grad_scaling=False, grad_scaler=2.5), dt)
        add("pt_gradscaler_" + dt, (2, 3, 5, 7), P(grad_scaler=0.37), dt)
        add("pt_zp_clamped_" + dt, (3, 11), P(), dt, scale=0.01, shift=-5.0)       # -b/s = 500 > tmax
        add("pt_zp_clamped_lo_" + dt, (3, 11), P(), dt, scale=0.01, shift=5.0)     # -b/s < tmin
        add("pt_negscale_" + dt, (3, 11), P(), dt, scale=-0.03)
        add("pt_zeroscale_" + dt, (3, 11), P(), dt, scale=0.0, x_mean=0.0, x_std=1e-6)
        add("pt_tinyscale_" + dt, (3, 11), P(), dt, scale=1e-9, x_mean=0.0, x_std=1e-6)
        add("pt_special_" + dt, (0,), P(), dt, scale=1.0, special="edge")
        add("pt_special_sym_" + dt, (0,), P(**sym8), dt, scale=1.0, special="edge")
        add("pt_special_init_" + dt, (0,), P(init_mode=True), dt, scale=1.0, special="edge")
        add("pt_ties_" + dt, (0,), P(), dt, scale=0.5, shift=0.25, special="ties")
        for n in (1, 3, 5, 255, 1023, 4099):
            add("pt_len%d_%s" % (n, dt), (n,), P(), dt)
        add("pc_axis1_affine_" + dt, (4, 8, 6, 6), P(is_perchannel=True, quant_min=-8, quant_max=7,
                                                       type_min=-128, type_max=127), dt,
            scale=(0.05, 0.35), shift=("normal", 0.0, 0.1), x_mean=0.0)
        add("pc_axis0_sym_" + dt, (8, 4, 3, 3), P(is_perchannel=True, axis=0, **sym8), dt,
            scale=(5e-4, 2.5e-3), x_mean=0.0, x_std=0.05)
        add("pc_axis0_sym7_" + dt, (16, 5, 3, 3), P(is_perchannel=True, axis=0, **sym7), dt,
            scale=(5e-4, 2.5e-3), x_mean=0.0, x_std=0.05)
        add("pc_lastaxis_" + dt, (5, 16), P(is_perchannel=True, axis=1), dt, scale=(0.01, 0.05),
            shift=("normal", 0.0, 0.05))
        add("pc_axis2_of3_" + dt, (3, 5, 6), P(is_perchannel=True, axis=2), dt, scale=(0.01, 0.05))
        add("pc_mid_inner7_" + dt, (3, 5, 7), P(is_perchannel=True, axis=1), dt, scale=(0.01, 0.05),
            shift=("normal", 0.0, 0.05))
        add("pc_eval_" + dt, (4, 8, 6, 6), P(is_perchannel=True, eval_mode=True), dt, scale=(0.01, 0.05))
        add("pc_init_" + dt, (4, 8, 6, 6), P(is_perchannel=True, init_mode=True), dt, scale=(0.01, 0.05),
            shift=("normal", 0.0, 0.05))
        add("pc_noscaling_" + dt, (4, 8, 6, 6), P(is_perchannel=True, use_grad_scaling=False, grad_scaler=3.0),
            dt, scale=(0.01, 0.05))
        add("pc_negzero_scale_" + dt, (2, 4, 9), P(is_perchannel=True), dt, scale="negzero")
        add("pc_repeat_scale_" + dt, (4, 8, 6, 6), P(is_perchannel=True), dt, scale=0.03,
            shift=("normal", 0.0, 0.05), nscale=1)       # size-1 scale, [C] shift (lsq.cpp:124-126)
        add("pc_repeat_shift_" + dt, (4, 8, 6, 6), P(is_perchannel=True), dt, scale=(0.01, 0.05),
            shift=0.02, nshift=1)
        add("pc_special_" + dt, (0,), P(is_perchannel=True, axis=0), dt, scale=1.0, special="edge_pc")
    return specs


def edge_vector(dtype):
    fi = np.finfo(dtype)
    v = [0.5, 1.5, 2.5, 3.5, -0.5, 126.5, 127.5, 200.0, -3.0, np.nan, np.inf, -np.inf, 0.0, -0.0,
         126.49999, 127.00001, 1e30, -1e30, fi.tiny, -fi.tiny, fi.tiny / 4, fi.max, -fi.max, 63.5, -64.5,
         -128.5, -127.5, 254.5, 255.5, 0.49999997, 1.0, 127.0, 126.99999, 1e-8, -1e-8]
    return np.array(v, dtype=dtype)


def ties_vector(dtype):
    # x*inv_s + zp lands on k + 0.5 for many k (scale 0.5, shift 0.25 -> zp = rne(-0.5) = -0)
    k = np.arange(-6, 300, dtype=np.float64)
    return np.concatenate([(k + 0.5) * 0.5, k * 0.5, (k + 0.25) * 0.5]).astype(dtype)


def build_small_inputs(spec, idx):
    dt = getattr(torch, spec["dtype"])
    npdt = np.dtype(spec["dtype"])
    p = spec["p"]
    if spec["special"] in ("edge", "edge_pc"):
        xv = edge_vector(npdt)
        if spec["special"] == "edge_pc":
            x = torch.from_numpy(np.stack([xv, xv * 0.5, -xv]))      # [3, n], axis 0
        else:
            x = torch.from_numpy(xv)
        g = S.normal_like(x.numel(), 1000 + idx, 0.0, 1.0, dtype=dt).view(x.shape)
    elif spec["special"] == "ties":
        x = torch.from_numpy(ties_vector(npdt))
        g = S.normal_like(x.numel(), 1000 + idx, 0.0, 1.0, dtype=dt).view(x.shape)
    else:
        shape = tuple(spec["shape"])
        n = int(np.prod(shape))
        x = S.normal_like(n, 2000 + idx, spec["x_mean"], spec["x_std"], dtype=dt).view(shape)
        g = S.normal_like(n, 3000 + idx, 0.0, 1e-3, dtype=dt).view(shape)
    C = x.shape[p["axis"]] if p["is_perchannel"] else 1
    ns = spec["nscale"] or C
    nb = spec["nshift"] or C
    sc = spec["scale"]
    if sc == "negzero":
        base = S.uniform_like(ns, 4000 + idx, 0.01, 0.05, dtype=dt)
        base[0] = -base[0]
        base[1] = 0.0
        if ns > 2:
            base[2] = 1e-12
        scale = base
    elif isinstance(sc, (tuple, list)):
        scale = S.uniform_like(ns, 4000 + idx, sc[0], sc[1], dtype=dt)
    else:
        scale = torch.full((ns,), float(sc), dtype=dt)
    sh = spec["shift"]
    if isinstance(sh, (tuple, list)):
        shift = S.normal_like(nb, 5000 + idx, sh[1], sh[2], dtype=dt)
    else:
        shift = torch.full((nb,), float(sh), dtype=dt)
    return x, g, scale, shift


def scaler_chain_cases(ref):
    """One saturated element with grad 1 -> ds == fp(qmax * scaler) exactly: pins lsq_cpu.cpp:103,250."""
    out = []
    shapes = [(1,), (7,), (64,), (1000,), (4, 64, 56, 56), (3, 5, 7, 11), (128, 3, 17), (999983,), (1 << 20,),
              (33, 77, 13), (2, 2), (65537,), (12, 345, 67), (1 << 22,), (5, 999), (48, 48, 48)]
    for shape in shapes:
        n = int(np.prod(shape))
        for qmax in (1, 7, 15, 63, 127, 255):
            for gscale in (1.0, 0.1):
                for dt in (torch.float32, torch.float64):
                    x = torch.zeros(shape, dtype=dt)
                    x.view(-1)[n // 2] = 1e6
                    g = torch.zeros(shape, dtype=dt)
                    g.view(-1)[n // 2] = 1.0
                    s = torch.ones(1, dtype=dt)
                    b = torch.zeros(1, dtype=dt)
                    p = P(quant_max=qmax, type_max=255, grad_scaler=gscale)
                    _, _, ds, _ = ref_fwd_bwd(ref, x, g, s, b, p)
                    exp = O.grad_scaler_pt(n, qmax, True, gscale, dtype=x.numpy().dtype)
                    want = np.array([qmax], dtype=ds.dtype) * np.array([exp], dtype=ds.dtype)
                    assert bits_equal(ds, want), ("scaler chain (per-tensor)", shape, qmax, gscale, ds, want)
                    rec = dict(kind="pt", shape=list(shape), qmax=qmax, grad_scaler=gscale,
                               dtype=str(ds.dtype), ds_hex=ds.tobytes().hex())
                    out.append(rec)
                    if len(shape) >= 2:
                        for axis in (0, 1):
                            C = shape[axis]
                            sc = torch.ones(C, dtype=dt)
                            bc = torch.zeros(C, dtype=dt)
                            p2 = P(quant_max=qmax, type_max=255, grad_scaler=gscale, is_perchannel=True, axis=axis)
                            _, _, dsc, _ = ref_fwd_bwd(ref, x, g, sc, bc, p2)
                            exp = O.grad_scaler_pc(n, qmax, C, True, gscale, dtype=x.numpy().dtype)
                            want = np.array([qmax], dtype=dsc.dtype) * np.array([exp], dtype=dsc.dtype)
                            nz = dsc[dsc != 0]
                            assert nz.size == 1 and bits_equal(nz, want), ("scaler chain (per-channel)", shape, axis, qmax, nz, want)
                            out.append(dict(kind="pc", shape=list(shape), axis=axis, qmax=qmax, grad_scaler=gscale,
                                            dtype=str(dsc.dtype), ds_hex=nz.tobytes().hex()))
    return out


def make_small(ref):
    specs = small_case_specs()
    arrays = {}
    manifest = []
    for i, spec in enumerate(specs):
        x, g, scale, shift = build_small_inputs(spec, i)
        p = spec["p"]
        ref_out = ref_fwd_bwd(ref, x, g, scale, shift, p)
        orc_out = oracle_fwd_bwd(x, g, scale, shift, p)
        check_against_oracle(spec["name"], ref_out, orc_out)
        k = "c%03d_" % i
        arrays[k + "x"] = x.numpy()
        arrays[k + "g"] = g.numpy()
        arrays[k + "scale"] = scale.numpy()
        arrays[k + "shift"] = shift.numpy()
        for nm, a in zip(("y", "dx", "ds", "db"), ref_out):
            arrays[k + nm] = a
        arrays[k + "abs_ds"] = orc_out[6]
        arrays[k + "abs_db"] = orc_out[7]
        manifest.append(dict(key=k, name=spec["name"], params=p, dtype=spec["dtype"], shape=list(x.shape)))
    # empty tensor behaviour (lsq_cpu.cpp:76-78: backward returns (x, scale, shift) unchanged)
    xe = torch.zeros(0, 3)
    ye = ref.lsq_forward_per_tensor(xe, torch.ones(1), torch.zeros(1), 0, 127, 0, 255, True, 1.0, False, False, False)
    be = ref.lsq_backward_per_tensor(xe, xe, torch.full((1,), 0.5), torch.full((1,), 0.25), 0, 127, 0, 255, True, 1.0,
                                     False, False, False)
    empty = dict(fwd_shape=list(ye.shape), bwd_shapes=[list(t.shape) for t in be],
                 bwd_scale_passthrough=float(be[1][0]), bwd_shift_passthrough=float(be[2][0]))
    np.savez_compressed(os.path.join(HERE, "small_cases.npz"), **arrays)
    chain = scaler_chain_cases(ref)
    with open(os.path.join(HERE, "small_cases.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_golden.py", torch=torch.__version__,
                       reference="DeadAt0m/LSQFakeQuantize-PyTorch torchlsq 2.1 CPU csrc (oracle/_ref/libtorchlsq_ref_ops.so)",
                       cases=manifest, empty=empty, scaler_chain=chain), f, indent=1)
    print("small cases:", len(manifest), "scaler-chain records:", len(chain))


# ---------------------------------------------------------------------------------------------
# BASELINE-config digests
# ---------------------------------------------------------------------------------------------
def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def levels_from_y(y, scale, shift, p, shape):
    """integer levels recovered from the reference's dequantised output: q = rne(y/s) + zp."""
    y64 = y.astype(np.float64)
    if p["is_perchannel"]:
        bshape = [1] * len(shape)
        bshape[p["axis"]] = shape[p["axis"]]
        s = np.maximum(np.abs(scale.astype(np.float32)), np.finfo(np.float32).eps).reshape(bshape)
        b = shift.astype(np.float32).reshape(bshape)
    else:
        s = np.float32(max(abs(np.float32(scale[0])), np.finfo(np.float32).eps))
        b = np.float32(shift[0])
    inv_s = (np.float32(1) / s).astype(np.float32)
    zp = np.rint(np.minimum(np.float32(p["type_max"]), np.maximum(np.float32(p["type_min"]), (-b * inv_s).astype(np.float32))))
    q = np.rint(y64 / s.astype(np.float64)) + zp.astype(np.float64)
    return q.astype(np.int16)


def digest_config(ref, name, cfg_key, bf16=False, abs_grad=False):
    t0 = time.time()
    c = S.CONFIGS[cfg_key]
    p = S.op_kwargs(cfg_key)
    if bf16:
        xb, gb, scale, shift = S.make_inputs(cfg_key, dtype=torch.bfloat16, abs_grad=abs_grad)
        x, g = xb.float(), gb.float()
    else:
        x, g, scale, shift = S.make_inputs(cfg_key, dtype=torch.float32, abs_grad=abs_grad)
    in_sha = dict(x=sha(x.numpy()), g=sha(g.numpy()), scale=sha(scale.numpy()), shift=sha(shift.numpy()))
    y, dx, ds, db = ref_fwd_bwd(ref, x, g, scale, shift, p)
    oy, odx, ods, odb, dsw, dbw, ads, adb = oracle_fwd_bwd(x, g, scale, shift, p)
    check_against_oracle(name, (y, dx, ds, db), (oy, odx, ods, odb, dsw, dbw, ads, adb))
    q = levels_from_y(y, scale.numpy(), shift.numpy(), p, x.shape)
    if p["is_perchannel"]:
        outer, C, inner = O.axis_to_ocl(x.shape, p["axis"])
        oq = O.levels_pc(x.numpy(), scale.numpy(), shift.numpy(), outer, C, inner, p["quant_min"], p["quant_max"],
                         p["type_min"], p["type_max"])
    else:
        oq = O.levels_pt(x.numpy(), scale[0].item(), shift[0].item(), p["quant_min"], p["quant_max"], p["type_min"],
                         p["type_max"])
    assert np.array_equal(q.astype(np.int32).reshape(-1), oq.reshape(-1)), name + ": levels from reference y != oracle levels"
    hist = np.bincount((q.reshape(-1).astype(np.int64) - p["quant_min"]), minlength=p["quant_max"] - p["quant_min"] + 1)
    rec = dict(config=cfg_key, shape=list(x.shape), params=p, bf16_io=bf16, abs_grad=abs_grad, inputs_sha256=in_sha,
               y_sha256=sha(y), dx_sha256=sha(dx), levels_int16_sha256=sha(q), level_hist=hist.tolist(),
               ds=ds.astype(np.float64).tolist(), db=db.astype(np.float64).tolist(),
               ds_f32_hex=ds.tobytes().hex() if ds.size <= 4 else sha(ds),
               db_f32_hex=db.tobytes().hex() if db.size <= 4 else sha(db),
               oracle_ds_wide=dsw.tolist(), oracle_db_wide=dbw.tolist(),
               oracle_abs_ds=ads.tolist(), oracle_abs_db=adb.tolist())
    if bf16:
        yb = torch.from_numpy(y).to(torch.bfloat16).view(torch.int16).numpy()
        dxb = torch.from_numpy(dx).to(torch.bfloat16).view(torch.int16).numpy()
        rec["y_bf16_sha256"] = sha(yb)
        rec["dx_bf16_sha256"] = sha(dxb)
    print("  %-14s %6.1fs  ds[0]=%.9g db[0]=%.9g" % (name, time.time() - t0, ds[0], db[0]))
    return rec


def shard_digests(ref):
    """What every RANK of a batch-sharded run must hold (bench.py --gpus N verifies itself against this; tests/test_bench_cli.py).

    cfg4 (strong scaling): rank r of N in {2, 4, 8} owns rows [r * 1024 / N, (r + 1) * 1024 / N) of the [1024,1024,14,14] tensor.
      y / dx are elementwise, so a rank's outputs are the slices of the reference's outputs on the whole tensor: sha256 per shard.
      The reduced [ds, db] must equal the reference's on the whole tensor (config "cfg4" above); each rank's own contribution is
      recorded from the oracle run on the shard with the GLOBAL element count in the gradient scaler (lsq_cpu.cpp:103 uses
      x.numel() of the concatenated tensor), and the sum of those contributions is checked HERE against the reference's result.
    cfg2_weak (weak scaling): rank r owns a full BASELINE-config-2 tensor, rows [128 r, 128 (r + 1)) of a virtual [128 N,512,56,56]
      one (synth.make_inputs(first_index = r * numel)).  y / dx: the reference's own ops on the shard's inputs (elementwise: what
      they would be inside the concatenated tensor).  ds / db contributions per (N, r): the oracle with the global count."""
    out = {}
    # ---- cfg4: one reference run on the whole tensor, sliced
    p = S.op_kwargs("cfg4")
    c = S.CONFIGS["cfg4"]
    x, g, scale, shift = S.make_inputs("cfg4", dtype=torch.float32)
    y, dx, ds, db = ref_fwd_bwd(ref, x, g, scale, shift, p)
    rows = x.shape[0]
    rec = dict(shape=list(x.shape), y_sha256=sha(y), dx_sha256=sha(dx), ds=ds.astype(np.float64).tolist(), db=db.astype(np.float64).tolist(), by_world={})
    xn, gn = x.numpy(), g.numpy()
    for N in (2, 4, 8):
        per = rows // N
        shards, tot_ds, tot_db, tot_ads, tot_adb = [], 0.0, 0.0, 0.0, 0.0
        for r in range(N):
            sl = slice(r * per, (r + 1) * per)
            o = O.bwd_pt(gn[sl], xn[sl], scale[0].item(), shift[0].item(), p["quant_min"], p["quant_max"], p["type_min"], p["type_max"],
                         True, 1.0, False, False, False, numel_for_scaler=x.numel())
            assert bits_equal(o.dx.reshape(dx[sl].shape), dx[sl]), "cfg4 shard %d/%d: oracle dx differs from the reference's slice" % (r, N)
            shards.append(dict(rank=r, first_index=r * per * int(np.prod(x.shape[1:])), shape=[per] + list(x.shape[1:]),
                               y_sha256=sha(y[sl]), dx_sha256=sha(dx[sl]),
                               ds_wide=float(o.ds_wide[0]), db_wide=float(o.db_wide[0]), abs_ds=float(o.abs_ds[0]), abs_db=float(o.abs_db[0])))
            tot_ds += float(o.ds_wide[0]); tot_db += float(o.db_wide[0]); tot_ads += float(o.abs_ds[0]); tot_adb += float(o.abs_db[0])
        # the method itself, against the reference on the concatenated tensor: shard sums with the global count == its ds / db
        assert abs(tot_ds - float(ds[0])) <= 1e-6 * tot_ads and abs(tot_db - float(db[0])) <= 1e-6 * tot_adb, (N, tot_ds, ds, tot_db, db)
        rec["by_world"][str(N)] = dict(shards=shards, sum_ds_wide=tot_ds, sum_db_wide=tot_db, abs_ds=tot_ads, abs_db=tot_adb)
        print("  cfg4 shards N=%d: sum of shard sums %.12g / %.12g vs reference %.9g / %.9g" % (N, tot_ds, tot_db, ds[0], db[0]))
    out["cfg4"] = rec
    del x, g, y, dx, xn, gn
    # ---- cfg2, weak-scaled: eight shards of a virtual [1024,512,56,56] tensor
    p = S.op_kwargs("cfg2")
    n = int(np.prod(S.CONFIGS["cfg2"]["shape"]))
    shards, per_world = [], {str(N): [] for N in (1, 2, 4, 8)}
    for r in range(8):
        t0 = time.time()
        x, g, scale, shift = S.make_inputs("cfg2", dtype=torch.float32, first_index=r * n)
        y, dx, ds, db = ref_fwd_bwd(ref, x, g, scale, shift, p)
        shards.append(dict(rank=r, first_index=r * n, x_sha256=sha(x.numpy()), g_sha256=sha(g.numpy()), y_sha256=sha(y), dx_sha256=sha(dx),
                           ds_alone=ds.astype(np.float64).tolist(), db_alone=db.astype(np.float64).tolist()))
        for N in (1, 2, 4, 8):
            if r >= N:
                continue
            o = O.bwd_pt(g.numpy(), x.numpy(), scale[0].item(), shift[0].item(), p["quant_min"], p["quant_max"], p["type_min"], p["type_max"],
                         True, 1.0, False, False, False, numel_for_scaler=N * n)
            assert bits_equal(o.dx.reshape(dx.shape), dx), "cfg2 weak shard %d: oracle dx differs from the reference" % r
            per_world[str(N)].append(dict(rank=r, ds_wide=float(o.ds_wide[0]), db_wide=float(o.db_wide[0]), abs_ds=float(o.abs_ds[0]),
                                          abs_db=float(o.abs_db[0])))
            if N == 1:      # a world of one is the plain op: the reference's own result on this shard
                assert abs(o.ds_wide[0] - float(ds[0])) <= 1e-6 * o.abs_ds[0] and abs(o.db_wide[0] - float(db[0])) <= 1e-6 * o.abs_db[0]
        print("  cfg2 weak shard %d  %5.1fs  ds alone %.9g" % (r, time.time() - t0, ds[0]))
        del x, g, y, dx
    out["cfg2_weak"] = dict(shard_shape=list(S.CONFIGS["cfg2"]["shape"]), shards=shards,
                            by_world={N: dict(shards=v, sum_ds_wide=sum(s_["ds_wide"] for s_ in v), sum_db_wide=sum(s_["db_wide"] for s_ in v),
                                              abs_ds=sum(s_["abs_ds"] for s_ in v), abs_db=sum(s_["abs_db"] for s_ in v))
                                      for N, v in per_world.items() if N != "1"})
    path = os.path.join(HERE, "shard_digests.json")
    with open(path, "w") as f:
        json.dump(dict(generator="tests/golden/make_golden.py --shards", torch=torch.__version__,
                       note="per-rank expectations of the batch-sharded bench runs: y / dx sha256 from the reference CPU csrc (slices of its "
                            "outputs on the whole tensor for cfg4; its outputs on the shard's inputs for cfg2_weak), each rank's fp64 [ds, db] "
                            "contribution from oracle/lsq_oracle.c run with the GLOBAL element count in the gradient scaler (checked at "
                            "generation time: their sum == the reference's ds / db on the concatenated cfg4 tensor within 1e-6 sum|terms|)",
                       shards=out), f, indent=1)
    print("wrote", path)


def make_digests(ref, big, only=None):
    path = os.path.join(HERE, "config_digests.json")
    out = {}
    if os.path.isfile(path):
        with open(path) as f:
            out = json.load(f).get("configs", {})
    jobs = [("cfg1", "cfg1", False, False), ("cfg1_absgrad", "cfg1", False, True),
            ("cfg3", "cfg3", False, False), ("cfg3_absgrad", "cfg3", False, True),
            ("cfg5_fp32", "cfg5", False, False), ("cfg5_bf16", "cfg5", True, False),
            ("cfg5_absgrad", "cfg5", False, True), ("cfg5_bf16_absgrad", "cfg5", True, True),
            # |grad| x the sign of each element's d_scale factor: every d_scale term >= 0 (synth.ds_term_sign)
            ("cfg3_dspos", "cfg3", False, "dspos"), ("cfg5_dspos", "cfg5", False, "dspos"),
            ("cfg5_bf16_dspos", "cfg5", True, "dspos"),
            # shapes the shipped library's policy sends to owner windows / the row groups' fat workgroup and ring
            # (synth.CONFIGS; tests/test_shipped_binary_gpu.py): the reference's own lsq_backward_per_channel_impl
            # (lsq_cpu.cpp:197-294) on them
            ("own33_fp32", "own33", False, False), ("own33_bf16", "own33", True, False),
            ("own64_fp32", "own64", False, False), ("own64_bf16", "own64", True, False),
            ("own16_fp32", "own16", False, False), ("own32_bf16", "own32", True, False),
            ("vit_fp32", "vit", False, False), ("vit_bf16", "vit", True, False),
            ("rgring_fp32", "rgring", False, False)]
    if big:
        jobs += [("cfg2", "cfg2", False, False), ("cfg2_absgrad", "cfg2", False, True),
                 ("cfg4", "cfg4", False, False)]
    for name, key, bf16, absg in jobs:
        if only and name not in only:
            continue
        out[name] = digest_config(ref, name, key, bf16, absg)
    with open(path, "w") as f:
        json.dump(dict(generator="tests/golden/make_golden.py", torch=torch.__version__, threads=torch.get_num_threads(),
                       note="expected outputs are the reference CPU csrc's; oracle_* fields come from oracle/lsq_oracle.c "
                            "(fp64 sums of the per-element fp32 terms and sum|term|, used as tolerance scale)",
                       configs=out), f, indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also digest cfg2/cfg4 (205 M elements each)")
    ap.add_argument("--skip-small", action="store_true")
    ap.add_argument("--shards", action="store_true", help="ONLY tests/golden/shard_digests.json: what each rank of a batch-sharded run must hold")
    ap.add_argument("--only", default="", help="comma-separated digest names to (re)generate; the others are kept as they are")
    a = ap.parse_args()
    import warnings
    warnings.filterwarnings("ignore")
    ref = load_reference()
    if a.shards:
        shard_digests(ref)
        sys.exit(0)
    if not a.skip_small:
        make_small(ref)
    make_digests(ref, a.big, set(a.only.split(",")) if a.only else None)
