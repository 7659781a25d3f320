"""CPU: pin the oracle (oracle/lsq_oracle.c) against the reference.

  * golden vectors produced by the reference's real CPU op library (tests/golden/make_golden.py);
  * the reference's own scalar header compiled from /root/reference (oracle/_ref/liblsq_ref_scalar.so),
    when that build is present (build container; it also travels to the GPU box).
"""
import ctypes
import os

import numpy as np
import pytest

from helpers import assert_bits_equal, assert_reduction_close, sha
from oracle import lsq_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_oracle(arrays, case):
    k, p = case["key"], case["params"]
    x, g, scale, shift = (arrays[k + n] for n in ("x", "g", "scale", "shift"))
    sym = not p["is_affine"]
    if p["is_perchannel"]:
        outer, C, inner = O.axis_to_ocl(x.shape, p["axis"])
        sc = scale if scale.size == C else np.repeat(scale, C)
        sh = shift if shift.size == C else np.repeat(shift, C)
        y = O.fwd_pc(x, sc, sh, outer, C, inner, p["quant_min"], p["quant_max"], p["type_min"], p["type_max"], p["init_mode"])
        r = O.bwd_pc(g, x, sc, sh, outer, C, inner, p["quant_min"], p["quant_max"], p["type_min"], p["type_max"],
                     p["use_grad_scaling"], p["grad_scaler"], sym, p["eval_mode"], p["init_mode"])
        ds, db = r.ds_wide, r.db_wide
        if scale.size != C:
            ds = np.array([ds.sum()])
        if shift.size != C:
            db = np.array([db.sum()])
        return y, r.dx, ds, db
    y = O.fwd_pt(x, scale[0], shift[0], p["quant_min"], p["quant_max"], p["type_min"], p["type_max"], p["init_mode"])
    r = O.bwd_pt(g, x, scale[0], shift[0], p["quant_min"], p["quant_max"], p["type_min"], p["type_max"],
                 p["use_grad_scaling"], p["grad_scaler"], sym, p["eval_mode"], p["init_mode"])
    return y, r.dx, r.ds_wide, r.db_wide


def test_oracle_matches_reference_small_cases(small_cases):
    manifest, arrays = small_cases
    assert len(manifest["cases"]) >= 60
    for case in manifest["cases"]:
        k = case["key"]
        y, dx, ds, db = _run_oracle(arrays, case)
        assert_bits_equal(y, arrays[k + "y"], case["name"] + " y")
        assert_bits_equal(dx, arrays[k + "dx"], case["name"] + " dx")
        assert_reduction_close(ds, arrays[k + "ds"], arrays[k + "abs_ds"], case["name"] + " ds")
        assert_reduction_close(db, arrays[k + "db"], arrays[k + "abs_db"], case["name"] + " db")


def test_oracle_grad_scaler_chain(small_cases):
    """ds of a single saturated element with grad 1 is fp(qmax * scaler): pins lsq_cpu.cpp:103,250 bit-for-bit."""
    manifest, _ = small_cases
    recs = manifest["scaler_chain"]
    assert len(recs) >= 700
    for r in recs:
        dt = np.dtype(r["dtype"])
        n = int(np.prod(r["shape"]))
        if r["kind"] == "pt":
            s = O.grad_scaler_pt(n, r["qmax"], True, r["grad_scaler"], dtype=dt)
        else:
            s = O.grad_scaler_pc(n, r["qmax"], r["shape"][r["axis"]], True, r["grad_scaler"], dtype=dt)
        want = np.frombuffer(bytes.fromhex(r["ds_hex"]), dtype=dt)
        got = np.array([r["qmax"]], dtype=dt) * np.array([s], dtype=dt)
        assert got.tobytes() == want.tobytes(), r


def test_empty_tensor_contract(small_cases):
    e = small_cases[0]["empty"]
    assert e["fwd_shape"] == [0, 3]
    assert e["bwd_shapes"] == [[0, 3], [1], [1]]
    assert e["bwd_scale_passthrough"] == 0.5 and e["bwd_shift_passthrough"] == 0.25


@pytest.mark.parametrize("name", ["cfg1", "cfg1_absgrad", "cfg3", "cfg3_absgrad", "cfg3_dspos"])
def test_oracle_matches_reference_config_digests(config_digests, name):
    import torch
    from torchlsq import synth
    d = config_digests[name]
    x, g, scale, shift = synth.make_inputs(d["config"], dtype=torch.float32, abs_grad=d["abs_grad"])
    assert sha(x.numpy()) == d["inputs_sha256"]["x"], "synthetic input generator drifted"
    assert sha(g.numpy()) == d["inputs_sha256"]["g"]
    p = d["params"]
    sym = not p["is_affine"]
    if p["is_perchannel"]:
        outer, C, inner = O.axis_to_ocl(x.shape, p["axis"])
        y = O.fwd_pc(x.numpy(), scale.numpy(), shift.numpy(), outer, C, inner, p["quant_min"], p["quant_max"],
                     p["type_min"], p["type_max"])
        r = O.bwd_pc(g.numpy(), x.numpy(), scale.numpy(), shift.numpy(), outer, C, inner, p["quant_min"], p["quant_max"],
                     p["type_min"], p["type_max"], True, 1.0, sym)
        q = O.levels_pc(x.numpy(), scale.numpy(), shift.numpy(), outer, C, inner, p["quant_min"], p["quant_max"],
                        p["type_min"], p["type_max"])
    else:
        y = O.fwd_pt(x.numpy(), scale[0].item(), shift[0].item(), p["quant_min"], p["quant_max"], p["type_min"], p["type_max"])
        r = O.bwd_pt(g.numpy(), x.numpy(), scale[0].item(), shift[0].item(), p["quant_min"], p["quant_max"],
                     p["type_min"], p["type_max"], True, 1.0, sym)
        q = O.levels_pt(x.numpy(), scale[0].item(), shift[0].item(), p["quant_min"], p["quant_max"], p["type_min"], p["type_max"])
    assert sha(y) == d["y_sha256"]
    assert sha(r.dx) == d["dx_sha256"]
    assert sha(q.astype(np.int16)) == d["levels_int16_sha256"]
    assert_reduction_close(r.ds_wide, d["ds"], d["oracle_abs_ds"], name + " ds")
    assert_reduction_close(r.db_wide, d["db"], d["oracle_abs_db"], name + " db")


# ---- against the reference's scalar header, compiled where it lies -----------------------------
_REF_SCALAR = os.path.join(ROOT, "oracle", "_ref", "liblsq_ref_scalar.so")


@pytest.mark.skipif(not os.path.isfile(_REF_SCALAR), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_bitwise_vs_reference_scalar_header(dtype):
    ref = ctypes.CDLL(_REF_SCALAR)
    suf = "f32" if dtype == np.float32 else "f64"
    cT = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    vp = ctypes.c_void_p
    rng = np.random.default_rng(7)
    n = 200_003
    eps = np.finfo(dtype).eps
    for trial, (qmin, qmax, tmin, tmax, sym, init, evalm, scale0, shift0) in enumerate([
            (0, 127, 0, 255, 0, 0, 0, 0.03, 0.1), (-64, 63, -128, 127, 1, 0, 0, 0.02, -0.0), (0, 255, 0, 255, 0, 1, 0, 0.05, -0.7),
            (-8, 7, -128, 127, 0, 0, 1, 0.3, 0.2), (0, 15, -128, 127, 0, 0, 0, 1e-12, 0.0), (-128, 127, -128, 127, 1, 1, 0, -0.01, 0.0)]):
        x = (rng.standard_normal(n) * 2 + 0.5).astype(dtype)
        x[:7] = [np.nan, np.inf, -np.inf, 0.0, -0.0, np.finfo(dtype).tiny / 2, 1e30]
        g = rng.standard_normal(n).astype(dtype)
        s = dtype(max(abs(dtype(scale0)), eps))
        inv_s = dtype(1) / s
        gs = dtype(O.grad_scaler_pt(n, qmax, True, 1.0, dtype=dtype))
        # forward
        y_ref = np.empty_like(x)
        f = getattr(ref, "ref_fwd_pt_" + suf)
        f.argtypes = [vp, vp, ctypes.c_int64] + [cT] * 7 + [ctypes.c_int]
        f(x.ctypes.data, y_ref.ctypes.data, n, s, inv_s, dtype(shift0), qmin, qmax, tmin, tmax, init)
        assert_bits_equal(O.fwd_pt(x, scale0, shift0, qmin, qmax, tmin, tmax, init), y_ref, "fwd trial %d" % trial)
        # backward incl. the per-element ds/db buffers the reference materialises
        dx_ref, dsb, dbb = np.empty_like(x), np.empty_like(x), np.empty_like(x)
        f = getattr(ref, "ref_bwd_pt_" + suf)
        f.argtypes = [vp] * 5 + [ctypes.c_int64] + [cT] * 8 + [ctypes.c_int] * 3
        f(g.ctypes.data, x.ctypes.data, dx_ref.ctypes.data, dsb.ctypes.data, dbb.ctypes.data, n, s, inv_s, dtype(shift0),
          qmin, qmax, tmin, tmax, gs, sym, evalm, init)
        r = O.bwd_pt(g, x, scale0, shift0, qmin, qmax, tmin, tmax, True, 1.0, sym, evalm, init, want_buffers=True)
        assert_bits_equal(r.dx, dx_ref, "dx trial %d" % trial)
        assert_bits_equal(r.ds_buf, dsb, "ds_buffer trial %d" % trial)
        assert_bits_equal(r.db_buf, dbb, "db_buffer trial %d" % trial)
    # per-channel walk
    outer, C, inner = 5, 37, 11
    x = rng.standard_normal(outer * C * inner).astype(dtype)
    g = rng.standard_normal(x.size).astype(dtype)
    sc = rng.uniform(-0.1, 0.1, C).astype(dtype)
    sc[3] = 0
    sh = rng.standard_normal(C).astype(dtype) * dtype(0.05)
    gs = dtype(O.grad_scaler_pc(x.size, 7, C, True, 1.0, dtype=dtype))
    y_ref = np.empty_like(x)
    f = getattr(ref, "ref_fwd_pc_" + suf)
    f.argtypes = [vp, vp] + [ctypes.c_int64] * 3 + [vp, vp] + [cT] * 4 + [ctypes.c_int, cT]
    f(x.ctypes.data, y_ref.ctypes.data, outer, C, inner, sc.ctypes.data, sh.ctypes.data, -8, 7, -128, 127, 0, eps)
    assert_bits_equal(O.fwd_pc(x, sc, sh, outer, C, inner, -8, 7, -128, 127), y_ref, "fwd_pc")
    dx_ref, dsb, dbb = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    f = getattr(ref, "ref_bwd_pc_" + suf)
    f.argtypes = [vp] * 5 + [ctypes.c_int64] * 3 + [vp, vp] + [cT] * 5 + [ctypes.c_int] * 3 + [cT]
    f(g.ctypes.data, x.ctypes.data, dx_ref.ctypes.data, dsb.ctypes.data, dbb.ctypes.data, outer, C, inner, sc.ctypes.data,
      sh.ctypes.data, -8, 7, -128, 127, gs, 0, 0, 0, eps)
    r = O.bwd_pc(g, x, sc, sh, outer, C, inner, -8, 7, -128, 127, True, 1.0, False, want_buffers=True)
    assert_bits_equal(r.dx, dx_ref, "dx_pc")
    assert_bits_equal(r.ds_buf, dsb, "ds_buffer_pc")
    assert_bits_equal(r.db_buf, dbb, "db_buffer_pc")


def test_oracle_meanstd_pinned(traces):
    """The statistics restatement against (1) torch.mean / torch.std, which is what the reference module calls
    (observers.py:329-337), and (2) the scales the reference module itself produced for its weight scenarios
    (tests/golden/module_traces.json, generated by importing the reference in place)."""
    import torch
    from torchlsq import synth
    for shape, axis in (((16, 8, 3, 3), 0), ((4, 8, 6, 6), 1), ((33, 1000), 1), ((7,), None), ((1, 5, 1), 1)):
        n = int(np.prod(shape))
        x = synth.normal_like(n, 5, 0.7, 1.3, dtype=torch.float64).view(shape)
        if axis is None:
            mu, sd = O.meanstd(x.numpy(), 1, 1, n)
            wmu, wsd = x.mean().reshape(1), x.std().reshape(1)
        else:
            outer, C, inner = O.axis_to_ocl(shape, axis)
            mu, sd = O.meanstd(x.numpy(), outer, C, inner)
            dims = [d for d in range(len(shape)) if d != axis]
            wmu, wsd = torch.mean(x, dims), torch.std(x, dims)
        np.testing.assert_allclose(mu, wmu.numpy(), rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(sd, wsd.numpy(), rtol=1e-12, atol=0, equal_nan=True)
    checked = 0
    for name, t in traces["traces"].items():
        sc = t["scenario"]
        if sc["ctor"].get("otype") != "weight":
            continue
        shape = sc["shape"]
        n = int(np.prod(shape))
        w = synth.normal_like(n, 100, sc["x_mean"], sc["x_std"]).view(shape).numpy()     # the creating call's input
        per_channel = "per_channel" in sc["ctor"].get("qscheme", "")
        axis = sc["ctor"].get("ch_axis", 0)
        outer, C, inner = O.axis_to_ocl(tuple(shape), axis) if per_channel else (1, 1, n)
        mu, sd = O.meanstd(w, outer, C, inner)
        fin = t["final"]
        scale = O.sigma_init_scale(mu, sd, fin["quant_min"], fin["quant_max"])
        # (an observer STATISTIC, not an output of the op: the reference's scale comes from torch.mean / torch.std in fp32 --
        #  two passes, its own ~1e-6 of rounding -- this build's from one fp64 pass; north_star's 1e-6 is about y / dx / d_scale / d_shift)
        np.testing.assert_allclose(scale, np.asarray(t["calls"][0]["scale"], dtype=np.float64), rtol=2e-6, atol=0,
                                   err_msg=name)
        checked += 1
    assert checked >= 3
