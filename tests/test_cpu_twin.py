"""CPU: the product's kernels for host memory (liblsq_cpu.so, include/lsq_cpu.h) against the pinned oracle.

The twin is the counterpart of the reference's CPU dispatch (lsq_cpu.cpp:298-311) and is held to the same bars as the
HIP kernels: y and dx bit-exact with the reference-generated goldens and with the oracle, d_scale / d_shift within
1e-6 * sum|terms|; plus what is its own: results do not depend on the OpenMP thread count.
"""
import numpy as np
import pytest
import torch

from helpers import assert_bits_equal, assert_reduction_close, sha
from oracle import lsq_oracle as O


@pytest.fixture(scope="module")
def E():
    import torchlsq  # noqa: F401
    from torchlsq import extension
    assert extension._CPU_LIB is not None, extension.cpu_error_str
    return extension


def _run(E, x, g, scale, shift, p, wide=False):
    sym = not p["is_affine"]
    tail = (p["quant_min"], p["quant_max"], p["type_min"], p["type_max"], p["use_grad_scaling"], p["grad_scaler"], sym,
            p["eval_mode"], p["init_mode"])
    pc = p["is_perchannel"]
    y = E.cpu_forward(x, scale, shift, p["axis"], pc, *tail)
    out = E.cpu_backward(g, x, scale, shift, p["axis"], pc, *tail, want_wide=wide)
    return (y,) + tuple(out)


def test_small_cases_match_the_reference_goldens(E, small_cases):
    """all 70 reference-generated cases (NaN / inf / ties / denormals / tiny and negative scales / modes), through the
    registered ops and autograd on CPU tensors"""
    from torchlsq.functional import lsq
    manifest, arrays = small_cases
    for case in manifest["cases"]:
        k, p = case["key"], case["params"]
        x = torch.from_numpy(arrays[k + "x"].copy()).requires_grad_(True)
        s = torch.from_numpy(arrays[k + "scale"].copy()).requires_grad_(True)
        b = torch.from_numpy(arrays[k + "shift"].copy()).requires_grad_(True)
        g = torch.from_numpy(arrays[k + "g"].copy())
        y = lsq(x, s, b, quant_min=p["quant_min"], quant_max=p["quant_max"], type_min=p["type_min"], type_max=p["type_max"],
                axis=p["axis"], use_grad_scaling=p["use_grad_scaling"], grad_scaler=p["grad_scaler"], is_affine=p["is_affine"],
                is_perchannel=p["is_perchannel"], eval_mode=p["eval_mode"], init_mode=p["init_mode"])
        y.backward(g)
        assert_bits_equal(y.detach().numpy(), arrays[k + "y"], case["name"] + " y")
        assert_bits_equal(x.grad.numpy(), arrays[k + "dx"], case["name"] + " dx")
        ds = s.grad.numpy() if s.grad is not None else np.zeros_like(arrays[k + "ds"])
        db = b.grad.numpy() if b.grad is not None else np.zeros_like(arrays[k + "db"])
        assert_reduction_close(ds, arrays[k + "ds"], arrays[k + "abs_ds"], case["name"] + " ds")
        assert_reduction_close(db, arrays[k + "db"], arrays[k + "abs_db"], case["name"] + " db")


@pytest.mark.parametrize("cfg", ["cfg1", "cfg3"])
def test_baseline_configs_match_reference_digests(E, config_digests, cfg):
    """full-size BASELINE configurations 1 and 3: sha256 of y and dx as produced by the reference's own CPU library"""
    from torchlsq import synth
    d = config_digests[cfg]
    c = synth.CONFIGS[cfg]
    x, g, scale, shift = synth.make_inputs(cfg, dtype=torch.float32)
    p = dict(quant_min=c["qmin"], quant_max=c["qmax"], type_min=c["tmin"], type_max=c["tmax"], axis=c["axis"],
             use_grad_scaling=True, grad_scaler=1.0, is_affine=c["affine"], is_perchannel=c["per_channel"], eval_mode=False,
             init_mode=False)
    y, dx, ds, db = _run(E, x, g, scale, shift, p)
    assert sha(x.numpy()) == d["inputs_sha256"]["x"]
    assert sha(y.numpy()) == d["y_sha256"] and sha(dx.numpy()) == d["dx_sha256"]
    assert_reduction_close(ds.numpy(), d["ds"], d["oracle_abs_ds"], cfg + " ds")
    assert_reduction_close(db.numpy(), d["db"], d["oracle_abs_db"], cfg + " db")


def _oracle(x, g, scale, shift, p, shape):
    sym = not p["is_affine"]
    q = (p["quant_min"], p["quant_max"], p["type_min"], p["type_max"])
    if p["is_perchannel"]:
        outer, C, inner = O.axis_to_ocl(shape, p["axis"])
        y = O.fwd_pc(x, scale, shift, outer, C, inner, *q, p["init_mode"])
        r = O.bwd_pc(g, x, scale, shift, outer, C, inner, *q, p["use_grad_scaling"], p["grad_scaler"], sym, p["eval_mode"],
                     p["init_mode"])
    else:
        y = O.fwd_pt(x, scale[0], shift[0], *q, p["init_mode"])
        r = O.bwd_pt(g, x, scale[0], shift[0], *q, p["use_grad_scaling"], p["grad_scaler"], sym, p["eval_mode"], p["init_mode"])
    return y, r


@pytest.mark.parametrize("seed", range(24))
def test_random_cases_equal_the_oracle(E, seed):
    rng = np.random.RandomState(1000 + seed)
    dims = rng.randint(1, 5)
    shape = tuple(int(v) for v in rng.choice([1, 2, 3, 5, 7, 16, 33], size=dims))
    dt = [np.float32, np.float64][seed % 2]
    pc = bool(seed % 3)
    axis = int(rng.randint(0, dims))
    C = shape[axis] if pc else 1
    x = (rng.standard_normal(shape) * 2 + 0.5).astype(dt)
    g = (rng.standard_normal(shape) * 1e-2).astype(dt)
    if seed % 5 == 0 and x.size > 3:
        x.reshape(-1)[:3] = [np.nan, np.inf, -np.inf]
    scale = (rng.uniform(0.01, 0.5, C) * rng.choice([1, -1], C)).astype(dt)
    shift = (rng.standard_normal(C) * 0.2).astype(dt)
    p = dict(quant_min=int(rng.choice([0, -8, -128])), quant_max=int(rng.choice([7, 127, 255])), type_min=-128, type_max=255,
             axis=axis, use_grad_scaling=bool(seed % 2), grad_scaler=float(rng.choice([1.0, 0.37])), is_affine=bool(seed % 4),
             is_perchannel=pc, eval_mode=seed % 7 == 3, init_mode=seed % 11 == 5)
    if not p["is_affine"]:
        p["quant_min"] = min(p["quant_min"], 0)
    out = _run(E, torch.from_numpy(x), torch.from_numpy(g), torch.from_numpy(scale), torch.from_numpy(shift), p)
    oy, r = _oracle(x, g, scale, shift, p, shape)
    assert_bits_equal(out[0].numpy(), oy, "y")
    assert_bits_equal(out[1].numpy(), r.dx, "dx")
    assert_reduction_close(out[2].numpy(), r.ds_wide, r.abs_ds, "ds")
    assert_reduction_close(out[3].numpy(), r.db_wide, r.abs_db, "db")


def test_bf16_storage_is_fp32_math_rounded_once(E):
    from torchlsq import synth
    x = synth.normal_like(4 * 16 * 49, 3, 0.0, 1.0, dtype=torch.bfloat16).view(4, 16, 7, 7)
    g = synth.normal_like(4 * 16 * 49, 4, 0.0, 1e-3, dtype=torch.bfloat16).view(4, 16, 7, 7)
    scale, shift = synth.uniform_like(16, 5, 0.05, 0.35), synth.normal_like(16, 6, 0.0, 0.1)
    p = dict(quant_min=-8, quant_max=7, type_min=-128, type_max=127, axis=1, use_grad_scaling=True, grad_scaler=1.0,
             is_affine=True, is_perchannel=True, eval_mode=False, init_mode=False)
    y, dx, ds, db = _run(E, x, g, scale, shift, p)
    oy, r = _oracle(x.float().numpy(), g.float().numpy(), scale.numpy(), shift.numpy(), p, tuple(x.shape))
    assert torch.equal(y, torch.from_numpy(oy).to(torch.bfloat16).view(x.shape))
    assert torch.equal(dx, torch.from_numpy(r.dx).to(torch.bfloat16).view(x.shape))
    assert_reduction_close(ds.numpy(), r.ds_wide, r.abs_ds, "ds")


def test_results_do_not_depend_on_the_thread_count(E):
    from torchlsq import synth
    n = 3 * 40 * 1031
    x = synth.normal_like(n, 7, 0.5, 1.0).view(3, 40, 1031)
    g = synth.normal_like(n, 8, 0.0, 1e-3).view(3, 40, 1031)
    scale, shift = synth.uniform_like(40, 9, 0.02, 0.2), synth.normal_like(40, 10, 0.0, 0.1)
    ppc = dict(quant_min=-8, quant_max=7, type_min=-128, type_max=127, axis=1, use_grad_scaling=True, grad_scaler=1.0,
               is_affine=True, is_perchannel=True, eval_mode=False, init_mode=False)
    ppt = dict(ppc, is_perchannel=False)
    keep = torch.get_num_threads()
    res = []
    try:
        for k in (1, 3, 8):
            torch.set_num_threads(k)
            res.append([t.clone() for t in _run(E, x, g, scale, shift, ppc, wide=True)] +
                       [t.clone() for t in _run(E, x, g, scale[:1], shift[:1], ppt, wide=True)])
    finally:
        torch.set_num_threads(keep)
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert a.numpy().tobytes() == b.numpy().tobytes()


def test_empty_strided_and_error_behaviour(E):
    from torchlsq.functional import lsq
    s, b = torch.tensor([0.1]), torch.tensor([0.0])
    assert lsq(torch.empty(0, 4), s, b).shape == (0, 4)
    x = torch.randn(6, 10)
    assert torch.equal(lsq(x.t(), s, b), lsq(x.t().contiguous(), s, b)) and lsq(x.t(), s, b).stride() == x.t().stride()
    with pytest.raises(RuntimeError, match="must have the same floating-point type"):
        lsq(x, s.double(), b)
    with pytest.raises(RuntimeError, match="not implemented for 'float16'"):
        lsq(x.half(), s, b)
