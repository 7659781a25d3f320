"""GPU parity tests (run with -m gpu on the MI355X box).

Everything goes through the product path: torch.ops.torchlsq.* -> ctypes -> liblsq_hip.so (C ABI)
-> gfx950 kernels.  Expected values are the reference CPU csrc's (committed golden fixtures) and
the CPU oracle on the same seeded inputs.  Bars (BASELINE.json north_star):
    y, dx, integer levels : bit-exact
    d_scale, d_shift      : |got - ref| <= 1e-6 * sum|terms|   (= 1e-6 relative when terms do not cancel)
"""
import numpy as np
import pytest
import torch

from helpers import TOL, assert_bits_equal, assert_reduction_close, sha
from oracle import lsq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    import torchlsq  # noqa: F401
    from torchlsq import extension
    extension._assert_has_ops()   # no native library -> fail loudly, never fall back
    return torch.device("cuda:0")


@pytest.fixture(params=["native", "ctypes"])
def host_binding(request, dev):
    """Run the test once per host layer of functional.lsq: the C++ torch binding (_lsq_torch.so) and the
    Python/ctypes autograd.Function.  Both sit on the same C ABI; a missing C++ binding fails the test."""
    from torchlsq import extension
    before = extension.host_binding()
    extension.set_host_binding(request.param)
    assert extension.host_binding() == request.param
    yield request.param
    extension.set_host_binding(before)


def _lsq_fwd_bwd(dev, x, g, scale, shift, p):
    from torchlsq.functional import lsq
    xg = torch.from_numpy(x).to(dev).requires_grad_(True)
    sg = torch.from_numpy(scale).to(dev).requires_grad_(True)
    bg = torch.from_numpy(shift).to(dev).requires_grad_(True)
    y = lsq(xg, sg, bg, p["quant_min"], p["quant_max"], p["type_min"], p["type_max"], p["axis"],
            p["use_grad_scaling"], p["grad_scaler"], p["is_affine"], p["is_perchannel"], p["eval_mode"], p["init_mode"])
    y.backward(torch.from_numpy(g).to(dev))
    torch.cuda.synchronize()
    zs = lambda t, like: np.zeros_like(like) if t is None else t.cpu().numpy()
    return y.detach().cpu().numpy(), xg.grad.cpu().numpy(), zs(sg.grad, scale), zs(bg.grad, shift)


def test_small_cases_match_reference(dev, small_cases, host_binding):
    manifest, arrays = small_cases
    for case in manifest["cases"]:
        k, p = case["key"], case["params"]
        y, dx, ds, db = _lsq_fwd_bwd(dev, arrays[k + "x"], arrays[k + "g"], arrays[k + "scale"], arrays[k + "shift"], p)
        assert_bits_equal(y, arrays[k + "y"], case["name"] + " y")
        assert_bits_equal(dx, arrays[k + "dx"], case["name"] + " dx")
        assert_reduction_close(ds, arrays[k + "ds"], arrays[k + "abs_ds"], case["name"] + " ds")
        assert_reduction_close(db, arrays[k + "db"], arrays[k + "abs_db"], case["name"] + " db")


def test_grad_scaler_chain_on_device(dev, small_cases):
    """One saturated element with grad 1: ds must be fp(qmax * scaler) bit-for-bit (lsq_cpu.cpp:103,250)."""
    manifest, _ = small_cases
    ops = torch.ops.torchlsq
    for r in manifest["scaler_chain"][::7]:
        dt = torch.float32 if r["dtype"] == "float32" else torch.float64
        shape = tuple(r["shape"])
        n = int(np.prod(shape))
        x = torch.zeros(shape, dtype=dt, device=dev)
        g = torch.zeros(shape, dtype=dt, device=dev)
        x.view(-1)[n // 2] = 1e6
        g.view(-1)[n // 2] = 1.0
        want = np.frombuffer(bytes.fromhex(r["ds_hex"]), dtype=r["dtype"])
        if r["kind"] == "pt":
            _, ds, _ = ops.lsq_backward_per_tensor(g, x, torch.ones(1, dtype=dt, device=dev), torch.zeros(1, dtype=dt, device=dev),
                                                   0, r["qmax"], 0, 255, True, r["grad_scaler"], False, False, False)
            got = ds.cpu().numpy()
        else:
            C = shape[r["axis"]]
            _, ds, _ = ops.lsq_backward_per_channel(g, x, torch.ones(C, dtype=dt, device=dev), torch.zeros(C, dtype=dt, device=dev),
                                                    r["axis"], 0, r["qmax"], 0, 255, True, r["grad_scaler"], False, False, False)
            got = ds.cpu().numpy()
            got = got[got != 0]
        assert got.tobytes() == want.tobytes(), (r, got, want)


def _run_config(dev, d, dtype=torch.float32):
    """Regenerate the seeded inputs ON THE GPU (bit-identical to the CPU generator), run fwd+bwd."""
    from torchlsq import synth
    from torchlsq.functional import lsq
    x, g, scale, shift = synth.make_inputs(d["config"], device=dev, dtype=dtype, abs_grad=d["abs_grad"])
    x.requires_grad_(True)
    scale.requires_grad_(True)
    shift.requires_grad_(True)
    y = lsq(x, scale, shift, **synth.op_kwargs(d["config"]))
    y.backward(g)
    torch.cuda.synchronize()
    return x, g, scale, shift, y


def _sha_t(t):
    t = t.detach().contiguous()
    if t.dtype == torch.bfloat16:
        t = t.view(torch.int16)
    return sha(t.cpu().numpy())


# Where the sum has no cancellation, "within 1e-6 relative of the reference" is asserted as plain rtol = 1e-6:
#   d_scale: |grad| inputs of the per-tensor quint8 configs (zero point 0: the low border contributes nothing) and the
#            "dspos" inputs (|grad| x the sign of each element's d_scale factor, synth.ds_term_sign) of the per-channel ones --
#            with |grad| alone the two borders of a signed range still cancel (sum|terms| / |sum| up to 1e4 per channel, and
#            the reference's own fp32 at::sum is 6e-5 relative away from the exact sum there: profiles/r03_reduction_margin.txt);
#   d_shift: every |grad| input (its terms are the gradients of the saturated elements).
_STRICT_DS = ("cfg1_absgrad", "cfg2_absgrad", "cfg3_dspos", "cfg5_dspos", "cfg5_bf16_dspos")
_STRICT_DB = ("cfg1_absgrad", "cfg2_absgrad", "cfg5_absgrad", "cfg5_bf16_absgrad")


def _assert_reductions(name, d, scale, shift):
    ds = scale.grad.cpu().numpy()
    db = shift.grad.cpu().numpy() if shift.grad is not None else np.zeros(len(d["db"]))
    assert_reduction_close(ds, d["ds"], d["oracle_abs_ds"], name + " ds")
    assert_reduction_close(db, d["db"], d["oracle_abs_db"], name + " db")
    if name in _STRICT_DS:
        np.testing.assert_allclose(ds.astype(np.float64), np.array(d["ds"]), rtol=TOL, atol=0, err_msg=name + " ds, plain rtol")
    if name in _STRICT_DB:
        np.testing.assert_allclose(db.astype(np.float64), np.array(d["db"]), rtol=TOL, atol=0, err_msg=name + " db, plain rtol")


@pytest.mark.parametrize("name", ["cfg1", "cfg1_absgrad", "cfg3", "cfg3_absgrad", "cfg3_dspos", "cfg5_fp32", "cfg5_absgrad",
                                  "cfg5_dspos", "cfg2", "cfg2_absgrad", "cfg4"])
def test_baseline_configs_match_reference_digests(dev, config_digests, name):
    """BASELINE.json shapes at FULL size against digests of the reference CPU csrc's outputs."""
    d = config_digests[name]
    x, g, scale, shift, y = _run_config(dev, d)
    assert _sha_t(x) == d["inputs_sha256"]["x"], "GPU input generation is not bit-identical to the CPU generator"
    assert _sha_t(g) == d["inputs_sha256"]["g"]
    assert _sha_t(y) == d["y_sha256"], name + ": y differs from the reference"
    assert _sha_t(x.grad) == d["dx_sha256"], name + ": dx differs from the reference"
    _assert_reductions(name, d, scale, shift)


@pytest.mark.parametrize("name", ["cfg5_bf16", "cfg5_bf16_absgrad", "cfg5_bf16_dspos"])
def test_bf16_io_config5(dev, config_digests, name):
    """BASELINE config 5: bf16 in/out, fp32 math.  Definition: reference fp32 CPU csrc on
    the upcast input, y/dx rounded to bf16 (RNE); ds/db stay fp32."""
    d = config_digests[name]
    x, g, scale, shift, y = _run_config(dev, d, dtype=torch.bfloat16)
    assert y.dtype == torch.bfloat16 and x.grad.dtype == torch.bfloat16 and scale.grad.dtype == torch.float32
    assert _sha_t(x.float()) == d["inputs_sha256"]["x"]
    assert _sha_t(g.float()) == d["inputs_sha256"]["g"]
    assert _sha_t(y) == d["y_bf16_sha256"]
    assert _sha_t(x.grad) == d["dx_bf16_sha256"]
    _assert_reductions(name, d, scale, shift)


@pytest.mark.parametrize("name", ["cfg1", "cfg3", "cfg5_fp32", "cfg2"])
def test_integer_levels_bit_exact(dev, config_digests, name):
    """The quantized integer levels (int8 emission of the forward) against the reference-derived digest."""
    from torchlsq import synth
    d = config_digests[name]
    p = d["params"]
    x, _, scale, shift = synth.make_inputs(d["config"], device=dev, dtype=torch.float32)
    bias = 128 if p["quant_max"] > 127 else 0
    if p["is_perchannel"]:
        y, q = torch.ops.torchlsq.lsq_quantize_per_channel(x, scale, shift, p["axis"], p["quant_min"], p["quant_max"],
                                                           p["type_min"], p["type_max"], bias)
    else:
        y, q = torch.ops.torchlsq.lsq_quantize_per_tensor(x, scale, shift, p["quant_min"], p["quant_max"],
                                                          p["type_min"], p["type_max"], bias)
    assert q.dtype == torch.int8
    lv = q.to(torch.int16) + bias
    assert sha(lv.cpu().numpy()) == d["levels_int16_sha256"]
    hist = torch.bincount((lv.view(-1).to(torch.int64) - p["quant_min"]), minlength=p["quant_max"] - p["quant_min"] + 1)
    assert hist.cpu().tolist() == d["level_hist"]
    assert _sha_t(y) == d["y_sha256"]


# ---- layouts, alignment, ragged sizes -------------------------------------------------------------
def _oracle_pt(x, g, s, b, p, **kw):
    y = O.fwd_pt(x, s, b, p[0], p[1], p[2], p[3], kw.get("init_mode", False))
    r = O.bwd_pt(g, x, s, b, p[0], p[1], p[2], p[3], True, 1.0, kw.get("sym", False), kw.get("eval_mode", False),
                 kw.get("init_mode", False))
    return y, r


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 65537,
                               1048576 + 3, 3 * 1048576 + 1021])
def test_per_tensor_ragged_sizes(dev, n):
    from torchlsq import synth
    x = synth.normal_like(n, 5, 1.5, 1.0)
    g = synth.normal_like(n, 6, 0.0, 1e-3)
    p = (0, 127, 0, 255)
    ops = torch.ops.torchlsq
    s, b = torch.tensor([0.03], device=dev), torch.tensor([0.07], device=dev)
    y = ops.lsq_forward_per_tensor(x.to(dev), s, b, *p, True, 1.0, False, False, False)
    dx, ds, db = ops.lsq_backward_per_tensor(g.to(dev), x.to(dev), s, b, *p, True, 1.0, False, False, False)
    oy, r = _oracle_pt(x.numpy(), g.numpy(), 0.03, 0.07, p)
    assert_bits_equal(y.cpu().numpy(), oy, "y n=%d" % n)
    assert_bits_equal(dx.cpu().numpy(), r.dx, "dx n=%d" % n)
    assert_reduction_close(ds.cpu().numpy(), r.ds_wide, r.abs_ds, "ds n=%d" % n)
    assert_reduction_close(db.cpu().numpy(), r.db_wide, r.abs_db, "db n=%d" % n)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16, torch.float16])
def test_misaligned_views_take_the_scalar_path(dev, dtype):
    """x[1:] is not 16-byte aligned: the library must fall back to its element-wise kernels."""
    from torchlsq import synth
    n = 10007
    pdt = torch.float64 if dtype == torch.float64 else torch.float32
    xb = synth.normal_like(n + 1, 5, 1.5, 1.0, dtype=dtype)
    gb = synth.normal_like(n + 1, 6, 0.0, 1e-3, dtype=dtype)
    x, g = xb.to(dev)[1:], gb.to(dev)[1:]
    assert x.data_ptr() % 16 != 0
    p = (0, 127, 0, 255)
    s, b = torch.tensor([0.03], device=dev, dtype=pdt), torch.tensor([0.07], device=dev, dtype=pdt)
    ops = torch.ops.torchlsq
    y = ops.lsq_forward_per_tensor(x, s, b, *p, True, 1.0, False, False, False)
    dx, ds, db = ops.lsq_backward_per_tensor(g, x, s, b, *p, True, 1.0, False, False, False)
    odt = np.float64 if dtype == torch.float64 else np.float32
    xn = xb[1:].to(pdt).numpy().astype(odt)
    gn = gb[1:].to(pdt).numpy().astype(odt)
    sv = s.cpu().numpy()[0]
    bv = b.cpu().numpy()[0]
    oy, r = _oracle_pt(xn, gn, sv, bv, p)
    want_y = torch.from_numpy(oy).to(dtype)
    want_dx = torch.from_numpy(r.dx).to(dtype)
    assert torch.equal(y.cpu(), want_y)
    assert torch.equal(dx.cpu(), want_dx)
    assert_reduction_close(ds.cpu().numpy(), r.ds_wide, r.abs_ds, "ds")
    assert_reduction_close(db.cpu().numpy(), r.db_wide, r.abs_db, "db")


PC_SHAPES = [
    ((4, 8, 6, 6), 1), ((8, 4, 3, 3), 0), ((5, 16), 1), ((3, 5, 6), 2), ((3, 5, 7), 1), ((2, 3, 7, 7), 1),
    ((64, 64, 3, 3), 0), ((16, 256, 7, 7), 1), ((2, 2048, 7, 7), 1), ((33, 1000), 1), ((1000, 33), 0), ((7, 1031), 1),
    ((300, 16), 1), ((5000, 8), 1), ((129, 4096), 1), ((4, 3, 224, 224), 1), ((2, 512, 14, 14), 1), ((1, 1, 5), 1),
    ((6, 1, 9), 1), ((1, 7), 1), ((257, 3), 1), ((2, 6, 2), 1), ((9, 2050, 3), 1),
]


@pytest.mark.parametrize("shape,axis", PC_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_per_channel_layouts(dev, shape, axis, dtype):
    from torchlsq import synth
    n = int(np.prod(shape))
    C = shape[axis]
    x = synth.normal_like(n, 15, 0.0, 1.0, dtype=dtype).view(shape)
    g = synth.normal_like(n, 16, 0.0, 1e-3, dtype=dtype).view(shape)
    scale = synth.uniform_like(C, 17, 0.05, 0.35, dtype=dtype)
    shift = synth.normal_like(C, 18, 0.0, 0.1, dtype=dtype)
    p = (-8, 7, -128, 127)
    ops = torch.ops.torchlsq
    y = ops.lsq_forward_per_channel(x.to(dev), scale.to(dev), shift.to(dev), axis, *p, True, 1.0, False, False, False)
    dx, ds, db = ops.lsq_backward_per_channel(g.to(dev), x.to(dev), scale.to(dev), shift.to(dev), axis, *p, True, 1.0,
                                              False, False, False)
    outer, C_, inner = O.axis_to_ocl(shape, axis)
    oy = O.fwd_pc(x.numpy(), scale.numpy(), shift.numpy(), outer, C_, inner, *p)
    r = O.bwd_pc(g.numpy(), x.numpy(), scale.numpy(), shift.numpy(), outer, C_, inner, *p, True, 1.0, False)
    assert_bits_equal(y.cpu().numpy(), oy, "y")
    assert_bits_equal(dx.cpu().numpy(), r.dx, "dx")
    assert_reduction_close(ds.cpu().numpy(), r.ds_wide, r.abs_ds, "ds")
    assert_reduction_close(db.cpu().numpy(), r.db_wide, r.abs_db, "db")


def test_channels_last_and_permuted_inputs(dev):
    """The reference preserves the memory format (empty_like(..., Preserve), lsq_cpu.cpp:31,80) and
    accepts arbitrary strides; results must not depend on the layout."""
    from torchlsq import synth
    shape = (4, 16, 6, 10)
    n = int(np.prod(shape))
    x = synth.normal_like(n, 25, 0.0, 1.0).view(shape)
    g = synth.normal_like(n, 26, 0.0, 1e-3).view(shape)
    scale = synth.uniform_like(16, 27, 0.05, 0.35)
    shift = synth.normal_like(16, 28, 0.0, 0.1)
    p = (-8, 7, -128, 127)
    ops = torch.ops.torchlsq
    xd, gd, sd, bd = x.to(dev), g.to(dev), scale.to(dev), shift.to(dev)
    y0 = ops.lsq_forward_per_channel(xd, sd, bd, 1, *p, True, 1.0, False, False, False)
    dx0, ds0, db0 = ops.lsq_backward_per_channel(gd, xd, sd, bd, 1, *p, True, 1.0, False, False, False)
    outer, C, inner = O.axis_to_ocl(shape, 1)
    oy = O.fwd_pc(x.numpy(), scale.numpy(), shift.numpy(), outer, C, inner, *p)
    ref = O.bwd_pc(g.numpy(), x.numpy(), scale.numpy(), shift.numpy(), outer, C, inner, *p, True, 1.0, False)
    variants = {
        "channels_last": (xd.contiguous(memory_format=torch.channels_last), gd.contiguous(memory_format=torch.channels_last)),
        "grad_other_layout": (xd.contiguous(memory_format=torch.channels_last), gd),
        "permuted": (xd.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2), gd),
        "strided": (torch.cat([xd, xd], 3)[..., ::2][..., :10], gd),
    }
    for name, (xv, gv) in variants.items():
        if name == "strided":
            xv = torch.empty(4, 16, 6, 20, device=dev)[..., ::2]
            xv.copy_(xd)
            assert not xv.is_contiguous()
        y = ops.lsq_forward_per_channel(xv, sd, bd, 1, *p, True, 1.0, False, False, False)
        dx, ds, db = ops.lsq_backward_per_channel(gv, xv, sd, bd, 1, *p, True, 1.0, False, False, False)
        assert torch.equal(y, y0), name
        assert torch.equal(dx, dx0), name
        if name == "channels_last":
            assert y.is_contiguous(memory_format=torch.channels_last) and dx.is_contiguous(memory_format=torch.channels_last)
        # every layout against the ORACLE on the logical tensor (the reference's TensorIterator walks any strides in place,
        # lsq_cpu.cpp:229-249): y, dx bit-exact, the sums within 1e-6 * sum|terms| -- north_star's bar, not a looser one
        assert_bits_equal(y.contiguous().cpu().numpy(), oy.reshape(shape), "y " + name)
        assert_bits_equal(dx.contiguous().cpu().numpy(), ref.dx.reshape(shape), "dx " + name)
        assert_reduction_close(ds.cpu().numpy(), ref.ds_wide, ref.abs_ds, "ds " + name)
        assert_reduction_close(db.cpu().numpy(), ref.db_wide, ref.abs_db, "db " + name)
    # per-tensor on a channels-last tensor
    s1, b1 = torch.tensor([0.1], device=dev), torch.tensor([0.05], device=dev)
    ya = ops.lsq_forward_per_tensor(xd, s1, b1, *p, True, 1.0, False, False, False)
    yb = ops.lsq_forward_per_tensor(xd.contiguous(memory_format=torch.channels_last), s1, b1, *p, True, 1.0, False, False, False)
    assert torch.equal(ya, yb) and yb.is_contiguous(memory_format=torch.channels_last)


def test_modes_eval_init_sym(dev):
    from torchlsq import synth
    n = 300_001
    x = synth.normal_like(n, 35, 0.0, 1.0)
    g = synth.normal_like(n, 36, 0.0, 1e-3)
    ops = torch.ops.torchlsq
    s, b = torch.tensor([0.02], device=dev), torch.tensor([0.01], device=dev)
    p = (-64, 63, -128, 127)
    for sym in (False, True):
        for ev in (False, True):
            for init in (False, True):
                y = ops.lsq_forward_per_tensor(x.to(dev), s, b, *p, True, 1.0, sym, ev, init)
                dx, ds, db = ops.lsq_backward_per_tensor(g.to(dev), x.to(dev), s, b, *p, True, 1.0, sym, ev, init)
                oy, r = _oracle_pt(x.numpy(), g.numpy(), 0.02, 0.01, p, sym=sym, eval_mode=ev, init_mode=init)
                tag = "sym=%s eval=%s init=%s" % (sym, ev, init)
                assert_bits_equal(y.cpu().numpy(), oy, "y " + tag)
                assert_bits_equal(dx.cpu().numpy(), r.dx, "dx " + tag)
                assert_reduction_close(ds.cpu().numpy(), r.ds_wide, r.abs_ds, "ds " + tag)
                assert_reduction_close(db.cpu().numpy(), r.db_wide, r.abs_db, "db " + tag)
                if init:
                    assert torch.equal(y.cpu(), x) and torch.equal(dx.cpu(), g)
                if ev:
                    assert ds.item() == 0.0 and db.item() == 0.0


def test_empty_tensors(dev):
    ops = torch.ops.torchlsq
    xe = torch.zeros(0, 3, device=dev)
    s, b = torch.full((1,), 0.5, device=dev), torch.full((1,), 0.25, device=dev)
    y = ops.lsq_forward_per_tensor(xe, s, b, 0, 127, 0, 255, True, 1.0, False, False, False)
    assert y.shape == (0, 3)
    dx, ds, db = ops.lsq_backward_per_tensor(xe, xe, s, b, 0, 127, 0, 255, True, 1.0, False, False, False)
    assert dx.shape == (0, 3) and ds.item() == 0.5 and db.item() == 0.25      # lsq_cpu.cpp:76-78


def test_errors_match_reference_checks(dev, host_binding):
    from torchlsq.functional import lsq
    x = torch.randn(4, 8, device=dev)
    s, b = torch.ones(1, device=dev), torch.zeros(1, device=dev)
    with pytest.raises(RuntimeError, match="scale should be a 1-D tensor"):
        lsq(x, torch.tensor(1.0, device=dev), b)
    with pytest.raises(RuntimeError, match="shift should be a 1-D tensor"):
        lsq(x, s, torch.tensor(0.0, device=dev))
    with pytest.raises(RuntimeError, match="must have the same floating-point type"):
        lsq(x, s.double(), b)
    with pytest.raises(RuntimeError, match="not consistent with input tensor"):
        lsq(x, torch.ones(3, device=dev), torch.zeros(3, device=dev), is_perchannel=True)
    with pytest.raises(AssertionError):
        lsq(x, s, b, quant_min=1, quant_max=5, is_affine=False)
    y_cpu = lsq(x.cpu(), s.cpu(), b.cpu())   # CPU tensors -> the CPU kernels (liblsq_cpu.so), same bits as the GPU's
    assert y_cpu.device.type == "cpu" and torch.equal(y_cpu, lsq(x, s, b).cpu())
    with pytest.raises(RuntimeError, match="expected a tensor on the GPU|expected all tensors on"):
        lsq(x, s.cpu(), b)                   # devices are never mixed or substituted


def test_backward_is_deterministic_and_graph_safe(dev):
    """No atomics in the per-tensor reduction: two runs are bit-identical; no host sync in the op
    (scale/shift are read on the device), so it captures into a HIP graph."""
    from torchlsq import synth
    n = 8 * 1024 * 1024 + 5
    x = synth.normal_like(n, 45, 1.5, 1.0, device=dev)
    g = synth.normal_like(n, 46, 0.0, 1e-3, device=dev)
    s, b = torch.tensor([0.03], device=dev), torch.tensor([0.0], device=dev)
    ops = torch.ops.torchlsq
    a = ops.lsq_backward_per_tensor(g, x, s, b, 0, 127, 0, 255, True, 1.0, False, False, False)
    c = ops.lsq_backward_per_tensor(g, x, s, b, 0, 127, 0, 255, True, 1.0, False, False, False)
    assert all(torch.equal(u, v) for u, v in zip(a, c))
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ops.lsq_forward_per_tensor(x, s, b, 0, 127, 0, 255, True, 1.0, False, False, False)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            yg = ops.lsq_forward_per_tensor(x, s, b, 0, 127, 0, 255, True, 1.0, False, False, False)
            dg = ops.lsq_backward_per_tensor(g, x, s, b, 0, 127, 0, 255, True, 1.0, False, False, False)
        s.fill_(0.05)          # the replay must see the NEW scale: nothing was baked in on the host
        graph.replay()
    torch.cuda.synchronize()
    y2 = ops.lsq_forward_per_tensor(x, s, b, 0, 127, 0, 255, True, 1.0, False, False, False)
    d2 = ops.lsq_backward_per_tensor(g, x, s, b, 0, 127, 0, 255, True, 1.0, False, False, False)
    assert torch.equal(yg, y2) and all(torch.equal(u, v) for u, v in zip(dg, d2))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float64, torch.float16])
def test_grad_in_another_dense_order_takes_the_tiled_pass(dev, dtype, host_binding):
    """A gradient whose dense memory order differs from x's by one transposition (contiguous NCHW grad for a channels-last x,
    the reverse, a 3-D [B, T, F] pair) is brought into x's order by lsq_hip_relayout -- bit for bit what Tensor.copy_ gives --
    and the backward then equals the same-layout call exactly.  Ragged tiles, extents below and above the 64 x 64 tile, H*W = 1."""
    from torchlsq import _abi, extension as E, synth
    lib = E.library()
    code = _abi._DTYPE_CODE[dtype]
    for shape in ((3, 70, 5, 9), (2, 130, 1, 1), (1, 64, 8, 8), (5, 7, 3, 300), (2, 2048, 7, 7), (4, 1, 6, 6), (2, 33, 129)):
        n = int(np.prod(shape))
        g = synth.normal_like(n, 91, 0.0, 1.0, device=dev, dtype=dtype).view(shape)
        if len(shape) == 4:
            x_like = torch.empty(shape, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
        else:
            x_like = torch.empty(shape[0], shape[2], shape[1], device=dev, dtype=dtype).permute(0, 2, 1)      # memory [B][F][T]
        for src, like in ((g, x_like), (g.clone().as_strided(shape, x_like.stride()).copy_(g), g)):
            abc = E._transposition(src, like)
            want = torch.empty_like(like).copy_(src)
            if src.stride() == like.stride() or E._physical_order(src) == E._physical_order(like):
                assert abc is None                # (size-1 dims: the two orders are one and the same memory)
                continue
            assert abc is not None and abc[0] * abc[1] * abc[2] == n, (shape, abc)
            got = E._like_layout(src, like)
            assert got.stride() == like.stride() and torch.equal(got, want), shape
            raw = torch.empty_like(like)
            assert lib.lsq_hip_relayout(code, src.data_ptr(), raw.data_ptr(), *abc, None) == 0
            torch.cuda.synchronize()
            assert torch.equal(raw, want), shape
    # ... and through the ops: channels-last x with a contiguous grad == both channels-last, bit for bit (dx, d_scale, d_shift)
    shape = (6, 96, 7, 7)
    n = int(np.prod(shape))
    x = synth.normal_like(n, 92, 0.1, 1.0, device=dev, dtype=dtype).view(shape).contiguous(memory_format=torch.channels_last)
    g = synth.normal_like(n, 93, 0.0, 1e-2, device=dev, dtype=dtype).view(shape)
    pd = torch.float64 if dtype == torch.float64 else torch.float32
    s, b = synth.uniform_like(96, 94, 0.05, 0.3, device=dev, dtype=pd), synth.normal_like(96, 95, 0.0, 0.1, device=dev, dtype=pd)
    ops = torch.ops.torchlsq_native if host_binding == "native" else torch.ops.torchlsq
    a = ops.lsq_backward_per_channel(g, x, s, b, 1, -8, 7, -128, 127, True, 1.0, False, False, False)
    c = ops.lsq_backward_per_channel(g.contiguous(memory_format=torch.channels_last), x, s, b, 1, -8, 7, -128, 127, True, 1.0, False, False, False)
    assert a[0].is_contiguous(memory_format=torch.channels_last) and all(torch.equal(u, v) for u, v in zip(a, c))
    assert lib.lsq_hip_relayout(code, None, None, 2, 2, 2, None) == -1 and lib.lsq_hip_relayout(code, None, None, 0, 2, 2, None) == 0


@pytest.mark.parametrize("shape,axis,dtype", [
    ((256, 2048, 7, 7), 1, torch.bfloat16),      # BASELINE config 5: 256-lane windows, two channels per lane, LDS ring
    ((256, 2048, 7, 7), 1, torch.float32),       # ... fp32
    ((48, 96, 28, 28), 1, torch.float32),        # 256-lane windows, one channel per lane
    ((40, 320, 3, 5), 1, torch.float16),         # inner 15: V channels per packet (CPL == V), slots shared by many lanes
    ((33, 2048, 7, 7), 1, torch.float32),        # owner windows
    ((64, 197, 768), 2, torch.bfloat16),         # row groups, fat workgroup
    ((8192, 1024), 1, torch.float32),            # row groups, register loops
    ((512, 512, 3, 3), 0, torch.float32),        # segment walk
])
def test_per_channel_backward_is_bit_reproducible(dev, shape, axis, dtype):
    """Every kernel family adds in a FIXED order: 50 launches of the same backward give the same bits -- dx, d_scale, d_shift and
    the un-rounded fp64 sums (`*_wide`), which is what the sharded path all-reduces (Reproducibility; until
    round 6 the 256-lane windows added their four waves' totals in arrival order)."""
    from torchlsq import synth
    n = 1
    for d_ in shape:
        n *= d_
    x = synth.normal_like(n, 61, 0.2, 1.0, device=dev, dtype=dtype).view(shape)
    g = synth.normal_like(n, 62, 0.0, 1e-2, device=dev, dtype=dtype).view(shape)
    C = shape[axis]
    s, b = synth.uniform_like(C, 63, 0.02, 0.2, device=dev), synth.normal_like(C, 64, 0.0, 0.1, device=dev)
    ops = torch.ops.torchlsq
    first = ops.lsq_backward_per_channel_wide(g, x, s, b, axis, -8, 7, -128, 127, True, 1.0, False, False, False, n)
    first = [t.clone() for t in first]
    assert first[1].dtype == torch.float64 and bool(torch.isfinite(first[1]).all())
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for rep in range(50):
        if rep % 5 == 0:
            scratch.fill_(rep)                     # other work in between: the waves do not arrive the same way twice
        again = ops.lsq_backward_per_channel_wide(g, x, s, b, axis, -8, 7, -128, 127, True, 1.0, False, False, False, n)
        assert torch.equal(again[0], first[0]), "dx differs on launch %d" % rep
        assert again[1].view(torch.int64).equal(first[1].view(torch.int64)), "the fp64 sums differ on launch %d" % rep
    dx, ds, db = ops.lsq_backward_per_channel(g, x, s, b, axis, -8, 7, -128, 127, True, 1.0, False, False, False)
    assert torch.equal(ds, first[1][0].to(torch.float32)) and torch.equal(db, first[1][1].to(torch.float32))


def test_full_size_properties_cfg2(dev):
    """Size-independent properties at BASELINE config 2 (205 M elements)."""
    from torchlsq import synth
    d = synth.CONFIGS["cfg2"]
    x, g, scale, shift = synth.make_inputs("cfg2", device=dev, dtype=torch.float32)
    ops = torch.ops.torchlsq
    p = (d["qmin"], d["qmax"], d["tmin"], d["tmax"])
    y = ops.lsq_forward_per_tensor(x, scale, shift, *p, True, 1.0, False, False, False)
    # idempotence: fake-quantising a fake-quantised tensor changes nothing
    y2 = ops.lsq_forward_per_tensor(y, scale, shift, *p, True, 1.0, False, False, False)
    assert torch.equal(y, y2)
    # every output is one of the (qmax-qmin+1) dequantised levels
    assert torch.unique(y).numel() <= d["qmax"] - d["qmin"] + 1
    # dx is grad where the input is strictly inside the range, 0 elsewhere
    dx, ds, db = ops.lsq_backward_per_tensor(g, x, scale, shift, *p, True, 1.0, False, False, False)
    inside = (dx != 0)
    assert torch.equal(dx[inside], g[inside])
    # linearity of the parameter gradients in grad: bwd(2g) == 2*bwd(g) exactly (power of two)
    dx_2, ds_2, db_2 = ops.lsq_backward_per_tensor(g * 2, x, scale, shift, *p, True, 1.0, False, False, False)
    assert torch.equal(dx_2, dx * 2) and torch.equal(ds_2, ds * 2) and torch.equal(db_2, db * 2)
    # shards add up: wide sums of two halves == wide sum of the whole (the multi-GPU contract)
    n = x.numel()
    h = x.shape[0] // 2
    _, w_all = ops.lsq_backward_per_tensor_wide(g, x, scale, shift, *p, True, 1.0, False, False, False, n)
    _, w_a = ops.lsq_backward_per_tensor_wide(g[:h], x[:h], scale, shift, *p, True, 1.0, False, False, False, n)
    _, w_b = ops.lsq_backward_per_tensor_wide(g[h:], x[h:], scale, shift, *p, True, 1.0, False, False, False, n)
    np.testing.assert_allclose((w_a + w_b).cpu().numpy(), w_all.cpu().numpy(), rtol=1e-12, atol=0)
    assert torch.equal(w_all.to(torch.float32)[0], ds[0])


def test_dispatcher_path_equals_direct_path(dev, small_cases, host_binding):
    """functional.lsq binds the kernels directly for GPU tensors; torch.ops.torchlsq.lsq goes through the
    dispatcher (front op -> register_autograd -> CUDA-key kernels).  Both must give identical results,
    including the size-1 `repeat` route of the per-channel front op (lsq.cpp:124-126)."""
    from torchlsq.functional import lsq
    manifest, arrays = small_cases
    picked = [c for c in manifest["cases"] if c["name"].endswith("float32") and
              c["name"].startswith(("pt_affine7", "pt_sym7", "pt_init", "pc_axis1_affine", "pc_repeat_scale", "pc_repeat_shift", "pc_axis0_sym_"))]
    assert len(picked) >= 6
    for case in picked:
        k, p = case["key"], case["params"]
        outs = []
        for route in ("direct", "dispatcher"):
            x = torch.from_numpy(arrays[k + "x"]).to(dev).requires_grad_(True)
            s = torch.from_numpy(arrays[k + "scale"]).to(dev).requires_grad_(True)
            b = torch.from_numpy(arrays[k + "shift"]).to(dev).requires_grad_(True)
            args = (p["quant_min"], p["quant_max"], p["type_min"], p["type_max"], p["axis"], p["use_grad_scaling"],
                    p["grad_scaler"], p["is_affine"], p["is_perchannel"], p["eval_mode"], p["init_mode"])
            y = lsq(x, s, b, *args) if route == "direct" else torch.ops.torchlsq.lsq(x, s, b, *args)
            y.backward(torch.from_numpy(arrays[k + "g"]).to(dev))
            outs.append((y.detach(), x.grad, s.grad, b.grad))
        for a, c in zip(*outs):
            assert (a is None and c is None) or torch.equal(a, c), case["name"]
        assert_bits_equal(outs[1][0].cpu().numpy(), arrays[k + "y"], case["name"] + " y (dispatcher)")
        assert_bits_equal(outs[1][1].cpu().numpy(), arrays[k + "dx"], case["name"] + " dx (dispatcher)")
    # double backward is refused on both routes
    x = torch.randn(64, device=dev, requires_grad=True)
    s, b = torch.ones(1, device=dev, requires_grad=True), torch.zeros(1, device=dev, requires_grad=True)
    y = lsq(x, s, b, 0, 127, 0, 255)
    (gx,) = torch.autograd.grad(y.sum(), x, create_graph=True)
    with pytest.raises(RuntimeError):
        gx.sum().backward()


def test_module_on_gpu_qat_step(dev):
    """LSQFakeQuantizer over the HIP kernels: one QAT-style step for an activation and a weight quantizer."""
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torchlsq.quantized import LSQFakeQuantizer
    act = LSQFakeQuantizer(MovingAverageMinMaxObserver, "activation", init_batches=1).to(dev)
    wq = LSQFakeQuantizer(MovingAveragePerChannelMinMaxObserver, "weight", dtype=torch.qint8,
                          qscheme=torch.per_channel_symmetric).to(dev)
    w = torch.nn.Parameter(torch.randn(32, 16, 3, 3, device=dev) * 0.05)
    x = torch.rand(8, 16, 12, 12, device=dev)
    for step in range(4):
        out = torch.nn.functional.conv2d(act(x), wq(w), padding=1)
        out.square().mean().backward()
    assert act.scale.is_cuda and act.scale.grad is not None and act.shift.grad is not None
    assert wq.scale.shape == (32,) and wq.scale.grad is not None and wq.shift.grad is None
    assert torch.isfinite(w.grad).all()
    assert int(act.current_batch[0]) == 2 and int(act.observer_enabled[0]) == 0
    # the quantised weight takes at most 2^7 distinct values per channel (7-bit default range)
    qw = wq(w).detach()
    assert all(torch.unique(qw[c]).numel() <= 128 for c in range(0, 32, 8))


@pytest.mark.slow
def test_more_than_2_pow_31_elements(dev):
    """64-bit indexing: a tensor with more than 2^31 elements must equal its two halves processed separately
    (each half < 2^31), bit-for-bit for y/dx and additively for the un-rounded reductions."""
    from torchlsq import synth
    ops = torch.ops.torchlsq
    half = (1 << 30) + 2048
    n = 2 * half + 5                       # > 2^31, ragged tail
    x = torch.empty(n, dtype=torch.bfloat16, device=dev)
    g = torch.empty(n, dtype=torch.bfloat16, device=dev)
    chunk = 1 << 27
    for lo in range(0, n, chunk):          # fill in pieces to keep temporaries small
        m = min(chunk, n - lo)
        x[lo:lo + m] = synth.normal_like(m, 900 + lo // chunk, 1.5, 1.0, device=dev, dtype=torch.bfloat16)
        g[lo:lo + m] = synth.normal_like(m, 950 + lo // chunk, 0.0, 1e-3, device=dev, dtype=torch.bfloat16)
    s, b = torch.tensor([0.03], device=dev), torch.tensor([0.01], device=dev)
    p = (0, 127, 0, 255)
    y = ops.lsq_forward_per_tensor(x, s, b, *p, True, 1.0, False, False, False)
    dx, wide = ops.lsq_backward_per_tensor_wide(g, x, s, b, *p, True, 1.0, False, False, False, n)
    parts = [(0, half), (half, n)]
    acc = torch.zeros(2, dtype=torch.float64, device=dev)
    for lo, hi in parts:
        yp = ops.lsq_forward_per_tensor(x[lo:hi], s, b, *p, True, 1.0, False, False, False)
        dxp, wp = ops.lsq_backward_per_tensor_wide(g[lo:hi], x[lo:hi], s, b, *p, True, 1.0, False, False, False, n)
        assert torch.equal(y[lo:hi], yp) and torch.equal(dx[lo:hi], dxp), "mismatch in [%d, %d)" % (lo, hi)
        acc += wp
        del yp, dxp
    np.testing.assert_allclose(acc.cpu().numpy(), wide.cpu().numpy(), rtol=1e-12)
    del y, dx
    # per-channel, window mode with > 2^31 elements: [outer, 64, 32] split along outer
    C, inner = 64, 32
    outer = n // (C * inner)
    m = outer * C * inner
    xv, gv = x[:m].view(outer, C, inner), g[:m].view(outer, C, inner)
    sc = synth.uniform_like(C, 17, 0.02, 0.05, device=dev)
    sh = synth.normal_like(C, 18, 0.0, 0.01, device=dev)
    y = ops.lsq_forward_per_channel(xv, sc, sh, 1, *p, True, 1.0, False, False, False)
    dx, wide = ops.lsq_backward_per_channel_wide(gv, xv, sc, sh, 1, *p, True, 1.0, False, False, False, m)
    ho = outer // 2
    acc = torch.zeros(2, C, dtype=torch.float64, device=dev)
    for lo, hi in ((0, ho), (ho, outer)):
        yp = ops.lsq_forward_per_channel(xv[lo:hi], sc, sh, 1, *p, True, 1.0, False, False, False)
        dxp, wp = ops.lsq_backward_per_channel_wide(gv[lo:hi], xv[lo:hi], sc, sh, 1, *p, True, 1.0, False, False, False, m)
        assert torch.equal(y[lo:hi], yp) and torch.equal(dx[lo:hi], dxp)
        acc += wp
        del yp, dxp
    # 16-bit storage adds up to 4 consecutive rows in fp32 before its fp64 accumulation, and which rows share a pre-sum
    # depends on where a workgroup's slab starts: whole tensor and halves agree on the scale of the sum of |terms|
    # (every |term| <= |g| * 128 * scaler; scaler = 1 / sqrt(m * 127 / C)), not to the last fp64 digit
    scaler = 1.0 / np.sqrt(float(m) * 127.0 / C)
    bound = gv.float().abs().sum(dim=(0, 2)).double().cpu().numpy() * 128.0 * scaler
    err = np.abs(acc.cpu().numpy() - wide.cpu().numpy())
    assert np.all(err <= 1e-7 * bound[None, :]), (err.max(), bound.min())


MM_SHAPES = [((7,), None), ((4099,), None), ((3, 1 << 20), None), ((4, 8, 6, 6), 1), ((8, 4, 3, 3), 0), ((5, 16), 1),
             ((3, 5, 7), 1), ((2, 2048, 7, 7), 1), ((33, 1000), 1), ((1000, 33), 0), ((300, 16), 1), ((129, 4096), 1),
             ((4, 3, 224, 224), 1), ((64, 64, 3, 3), 0), ((1, 7), 1), ((257, 3), 1), ((9, 2050, 3), 1), ((512, 4608), 0),
             # above the small-tensor geometry (>= 2 M elements): window mode with 1 / VEC channels per lane, folded rows
             ((16, 256, 56, 56), 1), ((8192, 1024), 1), ((40000, 96), 1), ((16, 197, 768), 2)]


@pytest.mark.parametrize("shape,axis", MM_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
def test_minmax_matches_torch_aminmax(dev, shape, axis, dtype):
    """One-pass observer statistics == torch.aminmax (CPU) exactly, incl. NaN propagation per channel."""
    from torchlsq import synth
    n = int(np.prod(shape))
    x = synth.normal_like(n, 61, 0.3, 2.0, dtype=dtype).view(shape)
    pdt = torch.float64 if dtype == torch.float64 else torch.float32
    for with_nan in (False, True):
        if with_nan:
            x = x.clone()
            x.view(-1)[n // 3] = float("nan")
        if axis is None:
            mn, mx = torch.ops.torchlsq.lsq_minmax_per_tensor(x.to(dev))
            wmn, wmx = torch.aminmax(x.to(pdt))
        else:
            mn, mx = torch.ops.torchlsq.lsq_minmax_per_channel(x.to(dev), axis)
            y = x.to(pdt).transpose(0, axis).flatten(1)
            wmn, wmx = torch.aminmax(y, dim=1)
        assert mn.dtype == pdt and mn.shape == wmn.shape
        assert torch.equal(mn.cpu().isnan(), wmn.isnan()) and torch.equal(mx.cpu().isnan(), wmx.isnan())
        ok = ~wmn.isnan()
        assert torch.equal(mn.cpu()[ok], wmn[ok]) and torch.equal(mx.cpu()[ok], wmx[ok])
    # a sliced (misaligned) view and a channels-last tensor
    if axis is None and n > 8:
        v = x.to(dev).view(-1)[1:]
        mn, mx = torch.ops.torchlsq.lsq_minmax_per_tensor(v)
        w = torch.aminmax(x.view(-1)[1:].to(pdt))
        assert (mn.cpu().isnan() and w[0].isnan()) or (mn.cpu() == w[0] and mx.cpu() == w[1])


def test_accelerated_observers_equal_stock_observers(dev):
    """LSQFakeQuantizer's observers on the GPU (one-pass kernel) track exactly what the stock torch observers
    compute, batch after batch, and produce the same qparams."""
    from torch.ao.quantization import observer as O
    from torchlsq import synth
    from torchlsq.quantized.modules import hip_observers as H
    for stock, kw, shape in ((O.MinMaxObserver, {}, (8, 16, 5, 5)), (O.MovingAverageMinMaxObserver, {}, (8, 16, 5, 5)),
                             (O.PerChannelMinMaxObserver, dict(ch_axis=1), (8, 16, 5, 5)),
                             (O.MovingAveragePerChannelMinMaxObserver, dict(ch_axis=0), (32, 16, 3, 3)),
                             (O.MovingAveragePerChannelMinMaxObserver, dict(ch_axis=1), (8, 16, 7, 7))):
        fast_cls = H.accelerated(stock)
        assert fast_cls is not stock and issubclass(fast_cls, stock)
        a, b = stock(**kw), fast_cls(**kw).to(dev)
        n = int(np.prod(shape))
        for step in range(4):
            x = synth.normal_like(n, 70 + step, 0.2 * step, 1.0 + 0.3 * step).view(shape)
            a(x)
            b(x.to(dev))
            assert torch.equal(a.min_val, b.min_val.cpu()) and torch.equal(a.max_val, b.max_val.cpu()), (stock.__name__, step)
        sa, za = a.calculate_qparams()
        sb, zb = b.cpu().calculate_qparams()          # same statistics -> same qparams (torch's own arithmetic, on the CPU)
        assert torch.equal(sa, sb) and torch.equal(za, zb)
        assert list(a.state_dict().keys()) == list(b.state_dict().keys())


@pytest.mark.parametrize("kind", ["act_pt_affine", "act_pt_symmetric", "act_pc_affine", "weightless_minmax", "custom_range"])
def test_fused_observer_tail_equals_the_reference_sequence(dev, kind):
    """Initialisation batches of an observer-driven quantizer on the GPU: the one-launch tail (observer state update +
    torch's qparams + scale / shift store, lsq_hip_observer_update) leaves the module in EXACTLY the state the reference
    sequence -- observer forward, calculate_qparams, _set_weights -- leaves it in, batch after batch, and (from the second
    observed batch on) without a single host synchronisation."""
    from torch.ao.quantization import observer as O
    from torchlsq import synth
    from torchlsq.quantized import LSQFakeQuantizer
    cfg = {
        "act_pt_affine": (O.MovingAverageMinMaxObserver, dict(), (8, 16, 9, 9)),
        "act_pt_symmetric": (O.MovingAverageMinMaxObserver, dict(qscheme=torch.per_tensor_symmetric, avoid_torch_overflow=False),
                             (8, 16, 9, 9)),       # (torch refuses reduce_range with symmetric quint8)
        "act_pc_affine": (O.MovingAveragePerChannelMinMaxObserver, dict(qscheme=torch.per_channel_affine), (8, 16, 9, 9)),
        "weightless_minmax": (O.MinMaxObserver, dict(avoid_torch_overflow=False), (4, 8, 33)),
        "custom_range": (O.PerChannelMinMaxObserver, dict(qscheme=torch.per_channel_symmetric, quant_min=0, quant_max=15,
                                                          avoid_torch_overflow=False), (8, 16, 9, 9)),
    }[kind]
    obs_cls, kw, shape = cfg
    n = int(np.prod(shape))

    def make(fused):
        m = LSQFakeQuantizer(obs_cls, "activation", init_batches=6, **kw).to(dev)
        m.fuse_observer_tail = fused
        return m

    a, b = make(True), make(False)
    for step in range(6):
        x = synth.normal_like(n, 300 + step, 0.3 * step - 0.4, 1.0 + 0.2 * step, device=dev).view(shape)
        if step == 2:                        # from here on the fused module must not touch the host
            torch.cuda.synchronize()
            torch.cuda.set_sync_debug_mode("error")
        try:
            ya = a(x)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        yb = b(x)
        if step == 0:
            continue                         # the creating call passes its input through
        oa, ob = a.activation_post_process, b.activation_post_process
        assert torch.equal(oa.min_val.reshape(-1), ob.min_val.reshape(-1)), (kind, step)
        assert torch.equal(oa.max_val.reshape(-1), ob.max_val.reshape(-1)), (kind, step)
        assert torch.equal(a.scale, b.scale) and torch.equal(a.shift, b.shift), (kind, step, a.scale, b.scale, a.shift, b.shift)
        assert torch.equal(ya, yb)
    assert a.state_dict().keys() == b.state_dict().keys()


def test_steady_state_forward_backward_never_synchronises(dev, host_binding):
    """After the init phase a quantizer call must not block on the device: the reference tests its state
    buffers with Python `if`s (4 device syncs per call on the GPU); here decisions read a host mirror and
    the ops read scale/shift on the device."""
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torchlsq.quantized import LSQFakeQuantizer
    act = LSQFakeQuantizer(MovingAverageMinMaxObserver, "activation", init_batches=1).to(dev)
    wq = LSQFakeQuantizer(MovingAveragePerChannelMinMaxObserver, "weight", dtype=torch.qint8,
                          qscheme=torch.per_channel_symmetric).to(dev)
    w = torch.nn.Parameter(torch.randn(32, 16, 3, 3, device=dev) * 0.05)
    x = torch.rand(8, 16, 12, 12, device=dev)
    for _ in range(4):                      # creation + init batches (these may synchronise)
        (act(x).sum() + wq(w).sum()).backward()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for _ in range(3):
            out = act(x).sum() + wq(w).sum()
            out.backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert int(act.current_batch[0]) == 2 and act._h["batch"] == 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
def test_eval_mode_masked_backward(dev, dtype, host_binding):
    """eval mode through functional.lsq: the forward saves a 1-byte inside mask instead of x and the backward is
    dx = grad * mask; results must equal the reference-style eval backward (op level, from x) bit for bit."""
    from torchlsq import synth
    from torchlsq.functional import lsq
    ops = torch.ops.torchlsq
    pdt = torch.float64 if dtype == torch.float64 else torch.float32
    for shape, pc, axis in (((4, 16, 7, 7), False, 1), ((4, 16, 7, 7), True, 1), ((32, 16, 3, 3), True, 0), ((10007,), False, 0)):
        n = int(np.prod(shape))
        x = synth.normal_like(n, 81, 0.5, 1.0, dtype=dtype, device=dev).view(shape)
        g = synth.normal_like(n, 82, 0.0, 1.0, dtype=dtype, device=dev).view(shape)
        g.view(-1)[3] = float("inf")                      # a real multiply: inf * 0 = NaN, like the reference
        C = shape[axis] if pc else 1
        s = synth.uniform_like(C, 83, 0.05, 0.2, dtype=pdt, device=dev).requires_grad_(True)
        b = synth.normal_like(C, 84, 0.0, 0.1, dtype=pdt, device=dev).requires_grad_(True)
        xr = x.clone().requires_grad_(True)
        y = lsq(xr, s, b, -8, 7, -128, 127, axis, True, 1.0, True, pc, eval_mode=True)
        assert y.grad_fn is not None
        if host_binding == "ctypes":      # a Python node exposes what it saved: the 1-byte mask, never x
            assert all(t.dtype == torch.int8 or t.numel() == C for t in y.grad_fn.saved_tensors)
        y.backward(g)
        if pc:
            y0 = ops.lsq_forward_per_channel(x, s.detach(), b.detach(), axis, -8, 7, -128, 127, True, 1.0, False, True, False)
            dx0, ds0, db0 = ops.lsq_backward_per_channel(g, x, s.detach(), b.detach(), axis, -8, 7, -128, 127, True, 1.0,
                                                         False, True, False)
        else:
            y0 = ops.lsq_forward_per_tensor(x, s.detach(), b.detach(), -8, 7, -128, 127, True, 1.0, False, True, False)
            dx0, ds0, db0 = ops.lsq_backward_per_tensor(g, x, s.detach(), b.detach(), -8, 7, -128, 127, True, 1.0, False, True, False)
        assert torch.equal(y.detach(), y0)
        a, c = xr.grad, dx0
        assert torch.equal(a.isnan(), c.isnan()) and torch.equal(a[~a.isnan()], c[~c.isnan()])
        assert torch.equal(s.grad, torch.zeros_like(s)) and torch.equal(b.grad, torch.zeros_like(b))
        assert float(ds0.abs().sum()) == 0.0 and float(db0.abs().sum()) == 0.0
    # channels-last input: the mask keeps the layout, a contiguous grad is re-laid
    x = torch.randn(2, 8, 5, 5, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    s1, b1 = torch.tensor([0.1], device=dev), torch.tensor([0.0], device=dev)
    y = lsq(x, s1, b1, 0, 15, 0, 255, eval_mode=True)
    y.backward(torch.ones(2, 8, 5, 5, device=dev))
    dx0, _, _ = ops.lsq_backward_per_tensor(torch.ones(2, 8, 5, 5, device=dev), x.detach(), s1, b1, 0, 15, 0, 255, True, 1.0,
                                            False, True, False)
    assert torch.equal(x.grad, dx0)


NATIVE_LAYOUTS = [((4, 16, 6, 10), 1), ((32, 16, 3, 3), 0), ((8, 5, 7), 2), ((3, 1, 9), 1), ((4099,), 0), ((2, 3, 4, 5, 6), 3)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16, torch.float16])
def test_native_binding_ops_equal_ctypes_ops(dev, dtype):
    """torch.ops.torchlsq_native.* (C++ binding, public C ABI) against torch.ops.torchlsq.* (Python/ctypes): same
    kernels, same launch geometry -> identical bits for y, dx (and the per-tensor reductions), for every layout the host layer
    has to translate into [outer, C, inner] (row-major, channels-last, permuted, non-dense, size-1 channel axis)."""
    from torchlsq import extension, synth
    extension.set_host_binding("native")
    nat, ops = torch.ops.torchlsq_native, torch.ops.torchlsq
    pdt = torch.float64 if dtype == torch.float64 else torch.float32
    p = (-8, 7, -128, 127)
    for shape, axis in NATIVE_LAYOUTS:
        n = int(np.prod(shape))
        x = synth.normal_like(n, 91, 0.3, 1.0, dtype=dtype, device=dev).view(shape)
        g = synth.normal_like(n, 92, 0.0, 1e-2, dtype=dtype, device=dev).view(shape)
        C = shape[axis]
        sc = synth.uniform_like(C, 93, 0.05, 0.3, dtype=pdt, device=dev)
        sh = synth.normal_like(C, 94, 0.0, 0.1, dtype=pdt, device=dev)
        layouts = {"row_major": (x, g)}
        if len(shape) == 4:
            layouts["channels_last"] = (x.contiguous(memory_format=torch.channels_last), g.contiguous(memory_format=torch.channels_last))
            layouts["grad_other_layout"] = (x.contiguous(memory_format=torch.channels_last), g)
        if len(shape) >= 3:
            perm = list(range(len(shape)))[::-1]
            inv = [perm.index(i) for i in range(len(shape))]
            layouts["permuted"] = (x.permute(perm).contiguous().permute(inv), g)
        wide = torch.empty(shape[:-1] + (2 * shape[-1],), dtype=dtype, device=dev)[..., ::2]
        wide.copy_(x)
        layouts["non_dense"] = (wide, g)
        for name, (xv, gv) in layouts.items():
            tag = "%s %s axis %d %s" % (dtype, shape, axis, name)
            for sym, ev, init in ((False, False, False), (True, False, False), (False, True, False), (False, False, True)):
                tail = p + (True, 1.0, sym, ev, init)
                ya = nat.lsq_forward_per_channel(xv, sc, sh, axis, *tail)
                yb = ops.lsq_forward_per_channel(xv, sc, sh, axis, *tail)
                assert torch.equal(ya, yb) and ya.stride() == yb.stride(), tag
                ra = nat.lsq_backward_per_channel(gv, xv, sc, sh, axis, *tail)
                rb = ops.lsq_backward_per_channel(gv, xv, sc, sh, axis, *tail)
                assert torch.equal(ra[0], rb[0]) and ra[0].stride() == rb[0].stride(), tag
                # per-channel sums: the four waves of a workgroup add their fp64 run totals into LDS in no fixed
                # order, so two launches agree to fp64 rounding (exactly, once rounded to fp32), not bit for bit in fp64
                for a, b in zip(ra[1:], rb[1:]):
                    if dtype == torch.float64:
                        torch.testing.assert_close(a, b, rtol=1e-10, atol=1e-15, msg=tag)
                    else:
                        assert torch.equal(a, b), tag
            tail = p + (True, 1.0, False, False, False)
            ya = nat.lsq_forward_per_tensor(xv, sc[:1], sh[:1], *tail)
            yb = ops.lsq_forward_per_tensor(xv, sc[:1], sh[:1], *tail)
            assert torch.equal(ya, yb) and ya.stride() == yb.stride(), tag
            for a, b in zip(nat.lsq_backward_per_tensor(gv, xv, sc[:1], sh[:1], *tail),
                            ops.lsq_backward_per_tensor(gv, xv, sc[:1], sh[:1], *tail)):
                assert torch.equal(a, b), tag
    # empty input, reference early-outs (lsq_cpu.cpp:76-78)
    xe = torch.zeros(0, 3, dtype=dtype, device=dev)
    s1, b1 = torch.full((1,), 0.5, dtype=pdt, device=dev), torch.full((1,), 0.25, dtype=pdt, device=dev)
    assert nat.lsq_forward_per_tensor(xe, s1, b1, 0, 127, 0, 255, True, 1.0, False, False, False).shape == (0, 3)
    dx, ds, db = nat.lsq_backward_per_tensor(xe, xe, s1, b1, 0, 127, 0, 255, True, 1.0, False, False, False)
    assert dx.shape == (0, 3) and ds.item() == 0.5 and db.item() == 0.25
    # rejected arguments carry the reference's messages
    with pytest.raises(RuntimeError, match="`axis` must be between 0 and number of dimensions of input"):
        nat.lsq_forward_per_channel(x, sc, sh, 7, *p, True, 1.0, False, False, False)
    with pytest.raises(RuntimeError, match="scale and shift need to have the same dimensions"):
        nat.lsq_forward_per_channel(x, sc, sh[:1], axis, *p, True, 1.0, False, False, False)
    with pytest.raises(RuntimeError, match="are not the same size"):
        nat.lsq_backward_per_tensor(g.view(-1)[:5], x, s1, b1, *p, True, 1.0, False, False, False)
    with pytest.raises(RuntimeError, match="does not fit a 32-bit integer"):
        nat.lsq_forward_per_tensor(x, s1, b1, 0, 2 ** 40, 0, 255, True, 1.0, False, False, False)
    with pytest.raises(RuntimeError, match="not implemented for"):
        nat.lsq_forward_per_tensor(x.to(torch.int32), s1, b1, 0, 127, 0, 255, True, 1.0, False, False, False)


def test_native_binding_runs_on_the_callers_stream_and_device(dev):
    """The C++ binding hands the CURRENT stream of the tensor's device to the C ABI: work enqueued on a side stream
    must be ordered after that stream's earlier work (a long fill) without any synchronisation in between."""
    from torchlsq import extension
    from torchlsq.functional import lsq
    extension.set_host_binding("native")
    side = torch.cuda.Stream(device=dev)
    s, b = torch.tensor([0.5], device=dev), torch.tensor([0.0], device=dev)
    x = torch.empty(1 << 26, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        x.fill_(3.2)                       # 256 MiB fill, then the op on the same stream
        xr = x.requires_grad_(True)
        y = lsq(xr, s, b, 0, 15, 0, 255)
        y.backward(torch.ones_like(y))
    side.synchronize()
    assert float(y.detach().min()) == 3.0 and float(y.detach().max()) == 3.0 and float(xr.grad.min()) == 1.0


@pytest.mark.parametrize("shape,axis", [((8, 6, 7, 7), 1), ((64, 5, 3), 1), ((16, 8, 4, 4), 1), ((6, 40, 25), 1), ((4, 6, 300), 1),
                                        ((12, 6, 3, 3), 0)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_non_finite_gradients_stay_in_their_channel(dev, shape, axis, dtype):
    """An inf / NaN upstream gradient poisons d_scale / d_shift of ITS channel only (the reference sums every channel
    separately, lsq_cpu.cpp:287-292).  Lanes whose packet straddles two channels keep two disjoint sums, so the
    finite neighbour must come out finite and within tolerance -- also at the packets right at a channel border."""
    from torchlsq import synth
    n = int(np.prod(shape))
    C = shape[axis]
    x = synth.normal_like(n, 71, 0.0, 1.0, dtype=dtype).view(shape).clone()
    g = synth.normal_like(n, 72, 0.0, 1e-3, dtype=dtype).view(shape).clone()
    outer, C_, inner = O.axis_to_ocl(shape, axis)
    g3 = g.view(outer, C_, inner)
    x3 = x.view(outer, C_, inner)
    g3[0, 1, inner - 1] = float("inf")          # last element of channel 1 (shares a packet with channel 2 when inner % V != 0)
    x3[0, 1, inner - 1] = 100.0                 # saturated: contributes to d_shift as well
    g3[outer - 1, 4, 0] = float("nan")          # first element of channel 4
    g3[outer // 2, 4, inner // 2] = float("-inf")
    scale = synth.uniform_like(C, 73, 0.05, 0.35, dtype=dtype)
    shift = synth.normal_like(C, 74, 0.0, 0.1, dtype=dtype)
    p = (-8, 7, -128, 127)
    dx, ds, db = torch.ops.torchlsq.lsq_backward_per_channel(g.to(dev), x.to(dev), scale.to(dev), shift.to(dev), axis, *p,
                                                              True, 1.0, False, False, False)
    r = O.bwd_pc(g.numpy(), x.numpy(), scale.numpy(), shift.numpy(), outer, C_, inner, *p, True, 1.0, False)
    assert_bits_equal(dx.cpu().numpy(), r.dx, "dx")
    ds, db = ds.cpu().numpy(), db.cpu().numpy()
    want_s, want_b = np.asarray(r.ds_wide, dtype=np.float64), np.asarray(r.db_wide, dtype=np.float64)
    bad_s, bad_b = ~np.isfinite(want_s), ~np.isfinite(want_b)
    assert bad_s[1] and bad_s[4] and bad_b[1] and not bad_s[[0, 2, 3]].any()
    assert (np.isfinite(ds) == ~bad_s).all() and (np.isfinite(db) == ~bad_b).all(), (ds, want_s, db, want_b)
    assert (np.isnan(ds) == np.isnan(want_s)).all() and (np.isnan(db) == np.isnan(want_b)).all()
    inf_s = np.isinf(want_s)
    assert (ds[inf_s] == want_s[inf_s]).all()
    ok = ~bad_s
    assert_reduction_close(ds[ok], want_s[ok], np.asarray(r.abs_ds)[ok], "ds (finite channels)")
    ok = ~bad_b
    assert_reduction_close(db[ok], want_b[ok], np.asarray(r.abs_db)[ok], "db (finite channels)")


@pytest.mark.parametrize("shape,axis", MM_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
def test_meanstd_matches_oracle(dev, shape, axis, dtype):
    """One-pass mean / unbiased std (the 3-sigma weight initialisation, reference observers.py:329-337) against the
    fp64 two-pass oracle: 1e-6 relative for fp32 results (they are the correctly rounded fp64 values up to the
    accumulation error), 1e-12 for fp64; also with a mean 1e5 sigma away from zero (a one-pass sum of squares
    without the pivot loses everything there) and with non-finite inputs (torch semantics)."""
    from torchlsq import synth
    n = int(np.prod(shape))
    pdt = torch.float64 if dtype == torch.float64 else torch.float32
    rtol = 1e-12 if dtype == torch.float64 else 1e-6
    for mean, std in ((0.3, 2.0), (1000.0, 0.01)):
        if dtype == torch.bfloat16 and mean == 1000.0:
            continue                                     # bf16 cannot represent that data
        x = synth.normal_like(n, 62, mean, std, dtype=dtype).view(shape)
        if axis is None:
            mu, sd = torch.ops.torchlsq.lsq_meanstd_per_tensor(x.to(dev))
            wmu, wsd = O.meanstd(x.to(torch.float64).numpy(), 1, 1, n)
            assert mu.shape == () and sd.shape == ()
        else:
            mu, sd = torch.ops.torchlsq.lsq_meanstd_per_channel(x.to(dev), axis)
            outer, C, inner = O.axis_to_ocl(shape, axis)
            wmu, wsd = O.meanstd(x.to(torch.float64).numpy(), outer, C, inner)
            assert mu.shape == (C,) and sd.shape == (C,)
        assert mu.dtype == pdt and sd.dtype == pdt
        np.testing.assert_allclose(mu.cpu().numpy().reshape(-1), wmu, rtol=rtol, atol=1e-7 * std)
        np.testing.assert_allclose(sd.cpu().numpy().reshape(-1), wsd, rtol=rtol, atol=0, equal_nan=True)
    # non-finite inputs propagate like torch.mean / torch.std (per channel)
    x = synth.normal_like(n, 63, 0.0, 1.0, dtype=dtype).view(shape).clone()
    x.view(-1)[0] = float("inf")             # also the pivot of channel 0
    x.view(-1)[n - 1] = float("nan")
    xr = x.to(pdt)
    if axis is None:
        mu, sd = torch.ops.torchlsq.lsq_meanstd_per_tensor(x.to(dev))
        wmu, wsd = xr.mean(), xr.std()
    else:
        mu, sd = torch.ops.torchlsq.lsq_meanstd_per_channel(x.to(dev), axis)
        dims = [d for d in range(len(shape)) if d != axis]
        wmu, wsd = torch.mean(xr, dims), torch.std(xr, dims)
    mu, sd, wmu, wsd = (t.reshape(-1).cpu().double() for t in (mu, sd, wmu, wsd))
    assert torch.equal(mu.isnan(), wmu.isnan()) and torch.equal(mu.isinf(), wmu.isinf()), (mu, wmu)
    assert torch.equal(sd.isnan(), wsd.isnan()), (sd, wsd)
    ok = ~(wmu.isnan() | wmu.isinf())
    np.testing.assert_allclose(mu[ok].numpy(), wmu[ok].numpy(), rtol=1e-5, atol=1e-6)
    # misaligned view (scalar path)
    if axis is None and n > 8:
        v = synth.normal_like(n, 64, 0.3, 2.0, dtype=dtype)
        mu, sd = torch.ops.torchlsq.lsq_meanstd_per_tensor(v.to(dev)[1:])
        wmu, wsd = O.meanstd(v[1:].to(torch.float64).numpy(), 1, 1, n - 1)
        np.testing.assert_allclose(mu.item(), wmu[0], rtol=rtol, atol=2e-7)
        np.testing.assert_allclose(sd.item(), wsd[0], rtol=rtol)


def test_sigma_init_on_gpu_matches_reference_traces(dev, traces):
    """The weight quantizer's creating call on a GPU weight: scale from the one-pass statistics kernel == the scale
    the reference module computed with torch.mean / torch.std (goldens generated from the reference itself)."""
    import importlib.util
    import os
    from torchlsq import synth
    from torchlsq.quantized import LSQFakeQuantizer
    spec = importlib.util.spec_from_file_location("make_module_traces", os.path.join(os.path.dirname(__file__), "golden",
                                                                                       "make_module_traces.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    from torch.ao.quantization import observer as obs_mod
    checked = 0
    for name, t in traces["traces"].items():
        sc = t["scenario"]
        if sc["ctor"].get("otype") != "weight":
            continue
        observer = getattr(obs_mod, sc["observer"]) if sc["observer"] else None
        m = LSQFakeQuantizer(observer, **drv.build_kwargs(sc["ctor"])).to(dev)
        m.train()
        n = int(np.prod(sc["shape"]))
        w = synth.normal_like(n, 100, sc["x_mean"], sc["x_std"]).view(sc["shape"]).to(dev)
        assert m(w) is w                                                 # the creating call passes its input through
        want = np.asarray(t["calls"][0]["scale"], dtype=np.float64)
        assert m.scale.is_cuda and m.scale.dtype == torch.float32
        # (an observer STATISTIC, not an output of the op: the reference's scale comes from torch.mean / torch.std in fp32 --
        #  two passes, its own ~1e-6 of rounding -- this build's from one fp64 pass; north_star's 1e-6 is about y / dx / d_scale / d_shift)
        np.testing.assert_allclose(m.scale.detach().cpu().numpy().astype(np.float64), want, rtol=2e-6, atol=0, err_msg=name)
        checked += 1
    assert checked >= 3


def test_concurrent_threads_and_streams(dev):
    """The C ABI is re-entrant (include/lsq_hip.h): four host threads, each on its own stream, hammer forward and
    backward of different problems at once (ctypes drops the GIL during the calls; the C++ binding runs its backward
    on autograd's worker thread anyway); every thread must reproduce its single-threaded results bit for bit."""
    import threading
    from torchlsq import synth
    from torchlsq.functional import lsq
    problems = []
    for k, (shape, pc, axis) in enumerate((((64, 32, 14, 14), False, 1), ((8, 256, 7, 7), True, 1), ((256, 64, 3, 3), True, 0),
                                           ((3, 1048576 + 17), False, 0))):
        n = int(np.prod(shape))
        C = shape[axis] if pc else 1
        x = synth.normal_like(n, 300 + k, 0.5, 1.0, device=dev).view(shape)
        g = synth.normal_like(n, 310 + k, 0.0, 1e-2, device=dev).view(shape)
        s = synth.uniform_like(C, 320 + k, 0.02, 0.1, device=dev)
        b = synth.normal_like(C, 330 + k, 0.0, 0.05, device=dev)
        problems.append((x, g, s, b, pc, axis))

    def run(prob):
        x, g, s, b, pc, axis = prob
        xr, sr, br = x.clone().requires_grad_(True), s.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = lsq(xr, sr, br, -8, 7, -128, 127, axis, True, 1.0, True, pc)
        y.backward(g)
        return y.detach(), xr.grad, sr.grad, br.grad

    want = [run(p) for p in problems]
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for _ in range(40):
                    got = run(problems[i])
                stream.synchronize()
                for a, w, what in zip(got, want[i], ("y", "dx", "ds", "db")):
                    # per-tensor and segment-mode sums have a fixed order; window-mode sums agree after fp32 rounding
                    if not torch.equal(a, w):
                        errors.append("thread %d: %s differs" % (i, what))
        except Exception as e:      # noqa: BLE001 -- reported below
            errors.append("thread %d: %r" % (i, e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(problems))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


LARGE_PC = [((32, 256, 56, 56), 1), ((8192, 1024), 1), ((64, 197, 768), 2), ((64, 56, 56, 64), 3), ((4, 8, 262144), 1),
            ((1, 3, 1000, 1250), 1), ((2048, 2304), 0), ((40000, 96), 1)]


@pytest.mark.parametrize("shape,axis", LARGE_PC)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_large_per_channel_shapes(dev, shape, axis, dtype):
    """Per-channel shapes above the small-tensor geometry (>= 2 M elements), one per decomposition: window mode with one /
    two / VEC channels per lane, rows folded into a tile, long rows split into many windows, segment mode with few
    and with many channels -- the launch geometries the small fixed cases and the fuzz test (<= 0.6 M elements) do
    not reach.  Same bars: y, dx bit-exact, d_scale / d_shift within 1e-6 of sum|terms| (16-bit storage: the fp32
    result rounded to the storage type)."""
    from torchlsq import synth
    n = int(np.prod(shape))
    C = shape[axis]
    x32 = synth.normal_like(n, 401, 0.4, 1.0, dtype=dtype).view(shape).to(torch.float32)     # values representable in `dtype`
    g32 = synth.normal_like(n, 402, 0.0, 1e-3, dtype=dtype).view(shape).to(torch.float32)
    scale = synth.uniform_like(C, 403, 0.02, 0.08)
    shift = synth.normal_like(C, 404, 0.0, 0.1)
    p = (0, 127, 0, 255)
    ops = torch.ops.torchlsq
    xd, gd = x32.to(dev).to(dtype), g32.to(dev).to(dtype)
    y = ops.lsq_forward_per_channel(xd, scale.to(dev), shift.to(dev), axis, *p, True, 1.0, False, False, False)
    dx, ds, db = ops.lsq_backward_per_channel(gd, xd, scale.to(dev), shift.to(dev), axis, *p, True, 1.0, False, False, False)
    outer, C_, inner = O.axis_to_ocl(shape, axis)
    oy = O.fwd_pc(x32.numpy(), scale.numpy(), shift.numpy(), outer, C_, inner, *p)
    r = O.bwd_pc(g32.numpy(), x32.numpy(), scale.numpy(), shift.numpy(), outer, C_, inner, *p, True, 1.0, False)
    if dtype == torch.float32:
        assert_bits_equal(y.cpu().numpy(), oy, "y")
        assert_bits_equal(dx.cpu().numpy(), r.dx, "dx")
    else:
        assert torch.equal(y.cpu().view(torch.int16), torch.from_numpy(oy).to(dtype).view(torch.int16)), "y"
        assert torch.equal(dx.cpu().view(torch.int16), torch.from_numpy(r.dx).to(dtype).view(torch.int16)), "dx"
    assert_reduction_close(ds.cpu().numpy(), r.ds_wide, r.abs_ds, "ds")
    assert_reduction_close(db.cpu().numpy(), r.db_wide, r.abs_db, "db")
