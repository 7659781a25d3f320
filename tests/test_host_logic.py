"""CPU: the host side of the drop-in surface (functional.lsq, the op layer, LSQFakeQuantizer).

The product has no CPU kernels; conftest plugs the CPU ORACLE in under the CPU dispatch key so that
the product's Python layer runs here exactly as it does over the HIP kernels on the GPU box.
Expected behaviour comes from traces captured from the reference's own module
(tests/golden/module_traces.json, made by tests/golden/make_module_traces.py).
"""
import importlib.util
import json
import os
import pickle

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _load_driver():
    spec = importlib.util.spec_from_file_location("_trace_driver", os.path.join(GOLDEN, "make_module_traces.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _close(a, b, what):
    if a is None or b is None:
        assert a is None and b is None, what
        return
    # wiring check on small mixed-sign sums: the reference's fp32 at::sum and the oracle's fp64 sum differ
    # by cancellation error; the arithmetic bar (1e-6 of sum|terms|) is enforced in test_oracle_pinned.py
    np.testing.assert_allclose(np.array(a, dtype=np.float64), np.array(b, dtype=np.float64), rtol=1e-4, atol=1e-8,
                               err_msg=what)


def test_module_state_machine_matches_reference_traces(oracle_cpu_backend, traces):
    from torchlsq.quantized import LSQFakeQuantizer
    drv = _load_driver()
    assert len(traces["traces"]) >= 13
    for name, t in traces["traces"].items():
        calls, final = drv.drive(LSQFakeQuantizer, t["scenario"])
        for got, want in zip(calls, t["calls"]):
            tag = "%s call %d" % (name, want["call"])
            for k in ("y_is_x", "y_sha", "dx_sha", "scale_requires_grad", "shift_requires_grad", "current_batch",
                      "observer_enabled", "fake_quant_enabled", "learning_enabled", "n_batches", "initialized"):
                assert got.get(k) == want.get(k), "%s: %s got %r want %r" % (tag, k, got.get(k), want.get(k))
            assert got["scale"] == want["scale"] and got["shift"] == want["shift"], tag + " parameters"
            _close(got["scale_grad"], want["scale_grad"], tag + " scale.grad")
            _close(got["shift_grad"], want["shift_grad"], tag + " shift.grad")
        for k in ("state_dict_keys", "quant_min", "quant_max", "ch_axis", "is_perchannel", "is_affine", "init_shift", "qparams"):
            assert final[k] == t["final"][k], "%s final %s: %r vs %r" % (name, k, final[k], t["final"][k])
        assert final["repr"] == t["final"]["repr"], name


def test_helpers_match_reference(traces):
    from torchlsq.quantized import LSQFakeQuantizer
    for shift, scale, dt, want in traces["extras"]["convert_shift_to_zp"]:
        got = int(LSQFakeQuantizer.convert_shift_to_zp(torch.tensor(shift), torch.tensor(scale), getattr(torch, dt)))
        assert got == want
    for ot, dt, lowbit, want in traces["extras"]["default_ranges"]:
        m = LSQFakeQuantizer(None, ot, dtype=getattr(torch, dt), init_mode="learnable", avoid_torch_overflow=lowbit,
                             qscheme=torch.per_tensor_symmetric if ot == "weight" else torch.per_tensor_affine)
        assert [m.quant_min, m.quant_max] == want


def test_constructor_assertions():
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver as Obs
    from torchlsq.quantized import LSQFakeQuantizer as Q
    with pytest.raises(AssertionError, match="only following modes available"):
        Q(Obs, "activation", init_mode="bogus")
    with pytest.raises(AssertionError, match="awaited Observer class"):
        Q(Obs(), "activation")
    with pytest.raises(AssertionError, match="otype must be on of"):
        Q(Obs, "bias")
    with pytest.raises(AssertionError, match="only symmetric scheme for weight"):
        Q(Obs, "weight", dtype=torch.qint8)
    with pytest.raises(AssertionError, match="requires `qint8` type for weights"):
        Q(None, "weight", qscheme=torch.per_tensor_symmetric, init_mode="learnable")
    with pytest.raises(AssertionError, match="requires `quint8` type for activation"):
        Q(Obs, "activation", dtype=torch.qint8)
    with pytest.raises(AssertionError, match="must include 0"):
        Q(Obs, "activation", quant_min=1, quant_max=5)
    with pytest.raises(AssertionError, match="not exceed the maximum bit range"):
        Q(Obs, "activation", quant_min=0, quant_max=255)        # 7-bit cap with avoid_torch_overflow
    assert Q(Obs, "activation", quant_min=0, quant_max=255, avoid_torch_overflow=False).quant_max == 255


def test_with_args_factory_and_qconfig(oracle_cpu_backend):
    """`with_args` raises NameError in the reference (missing `partial` import, observers.py:64); here it
    builds a picklable factory that QConfig / prepare_qat can instantiate per module."""
    from torch.ao.quantization import QConfig
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torchlsq.quantized import LSQFakeQuantizer
    act = LSQFakeQuantizer.with_args(observer=MovingAverageMinMaxObserver, otype="activation", init_batches=1)
    wgt = LSQFakeQuantizer.with_args(observer=MovingAveragePerChannelMinMaxObserver, otype="weight", dtype=torch.qint8,
                                     qscheme=torch.per_channel_symmetric).with_args(grad_scaler=0.5)
    a1, a2, w = act(), act(), wgt()
    assert a1 is not a2 and isinstance(w, LSQFakeQuantizer) and w.grad_scaler == 0.5 and w.ch_axis == 0
    qc = QConfig(activation=act, weight=wgt)
    assert isinstance(qc.activation(), LSQFakeQuantizer)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.ReLU())
    model.qconfig = qc
    torch.ao.quantization.prepare_qat(model.train(), inplace=True)
    assert isinstance(model[0].weight_fake_quant, LSQFakeQuantizer)
    x = torch.randn(2, 3, 8, 8)
    model(x)                      # creates the parameters (passes through)
    out = model(x)
    out.sum().backward()
    assert model[0].weight_fake_quant.scale.grad is not None and model[0].weight_fake_quant.scale.shape == (8,)


def test_state_dict_roundtrip(oracle_cpu_backend):
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver as Obs
    from torchlsq.quantized import LSQFakeQuantizer as Q
    a = Q(Obs, "activation", init_batches=1)
    x = torch.rand(4, 8) + 0.5
    for _ in range(3):
        a(x)
    sd = a.state_dict()
    assert list(sd.keys()) == ['scale', 'shift', 'fake_quant_enabled', 'observer_enabled', 'learning_enabled',
                               'current_batch', 'activation_post_process.eps', 'activation_post_process.min_val',
                               'activation_post_process.max_val']
    b = Q(Obs, "activation", init_batches=1)
    b(x)                                   # parameters exist only after the first call (reference behaviour)
    b.load_state_dict(sd)
    assert torch.equal(b.scale, a.scale) and int(b.current_batch[0]) == int(a.current_batch[0])
    assert torch.equal(a(x), b(x))


def _reference_state(traces, name):
    z = np.load(os.path.join(GOLDEN, "module_state_dicts.npz"))
    meta = traces["state_dicts"][name]
    return meta, {k: z[name + "/" + k] for k in meta["keys"]}


@pytest.mark.parametrize("name", ["act_observer_pt", "act_learnable_pt", "weight_pc_sym", "act_observer_pc", "act_fakequant_only"])
def test_state_dict_interchanges_with_the_reference_module(oracle_cpu_backend, traces, name):
    """Checkpoints interchange (reference observers.py:244-257): a state_dict() the REFERENCE module
    produced mid-scenario (tests/golden/module_state_dicts.npz, dumped by make_module_traces.py) loads into this module after
    its creating call, and the scenario then continues exactly as the reference's own trace; this module's state_dict() at
    the same point has the reference's keys, shapes, dtypes and values."""
    from torchlsq.quantized import LSQFakeQuantizer
    drv = _load_driver()
    t = traces["traces"][name]
    meta, ref_state = _reference_state(traces, name)
    k = meta["after_call"]
    # (1) the reference checkpoint drives this module
    calls, final = drv.drive(LSQFakeQuantizer, t["scenario"], resume=(k, ref_state))
    resumed = {c["call"]: c for c in calls}
    assert sorted(resumed) == [0] + list(range(k + 1, t["scenario"]["calls"]))
    for want in t["calls"][k + 1:]:
        got = resumed[want["call"]]
        tag = "%s resumed, call %d" % (name, want["call"])
        for key in ("y_is_x", "y_sha", "dx_sha", "scale_requires_grad", "shift_requires_grad", "current_batch",
                    "observer_enabled", "fake_quant_enabled", "learning_enabled", "initialized"):
            assert got.get(key) == want.get(key), "%s: %s got %r want %r" % (tag, key, got.get(key), want.get(key))
        assert got["scale"] == want["scale"] and got["shift"] == want["shift"], tag + " parameters"
        _close(got["scale_grad"], want["scale_grad"], tag + " scale.grad")
    assert final["qparams"] == t["final"]["qparams"] and final["state_dict_keys"] == t["final"]["state_dict_keys"]
    # (2) this module's own checkpoint at the same point is the reference's
    _, _, own = drv.drive(LSQFakeQuantizer, t["scenario"], dump_state_after=k)
    assert list(own.keys()) == meta["keys"]
    for key in meta["keys"]:
        assert str(own[key].dtype) == meta["dtypes"][key] and list(own[key].shape) == meta["shapes"][key], (name, key)
        assert own[key].tobytes() == ref_state[key].tobytes(), "%s: state_dict()[%r] differs from the reference's" % (name, key)


def test_two_calls_before_one_backward_match_the_reference(oracle_cpu_backend, traces):
    """the observer rewrites scale / shift in place at the second call; the first call's eval-mode backward then runs on the
    parameters as they are by then -- the reference's behaviour (lsq_autograd.cpp:46-73), pinned by its own module's trace"""
    from torchlsq.quantized import LSQFakeQuantizer
    drv = _load_driver()
    want = traces["extras"]["two_calls_one_backward"]
    got = drv.two_calls_one_backward(LSQFakeQuantizer)
    assert got == want


def test_apply_helpers():
    import torchlsq.quantized as TQ
    from torch.ao.quantization import FakeQuantize
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver as Obs
    act = TQ.LSQFakeQuantizer(Obs, "activation")
    wgt = TQ.LSQFakeQuantizer(Obs, "weight", dtype=torch.qint8, qscheme=torch.per_tensor_symmetric)
    fq = FakeQuantize()
    net = torch.nn.ModuleList([act, wgt, fq])
    net.apply(TQ.disable_fake_quant_on_act)
    assert int(act.fake_quant_enabled[0]) == 0 and int(wgt.fake_quant_enabled[0]) == 1 and int(fq.fake_quant_enabled[0]) == 0
    net.apply(TQ.enable_fake_quant)
    assert int(act.fake_quant_enabled[0]) == 1 and int(fq.fake_quant_enabled[0]) == 1
    wgt.enable_static_estimate()          # learning off -> the weight observer becomes meaningful
    assert int(wgt.observer_enabled[0]) == 1
    net.apply(TQ.disable_observer_on_weights)
    assert int(wgt.observer_enabled[0]) == 0 and int(act.observer_enabled[0]) == 1 and int(fq.observer_enabled[0]) == 0
    net.apply(TQ.enable_observer_on_weights)
    assert int(wgt.observer_enabled[0]) == 1
    net.apply(TQ.disable_observer)
    assert int(act.observer_enabled[0]) == 0


def test_functional_and_front_op(oracle_cpu_backend, small_cases):
    from torchlsq.functional import lsq
    manifest, arrays = small_cases
    x = torch.randn(4, 8)
    s, b = torch.ones(1), torch.zeros(1)
    with pytest.raises(RuntimeError, match="scale should be a 1-D tensor"):
        lsq(x, torch.tensor(1.0), b)
    with pytest.raises(RuntimeError, match="shift should be a 1-D tensor"):
        lsq(x, s, torch.zeros(1, 1))
    with pytest.raises(AssertionError, match="must be covered 0"):
        lsq(x, s, b, quant_min=1, quant_max=5, is_affine=False)
    with pytest.raises(RuntimeError, match="same floating-point type"):
        lsq(x, s.double(), b)
    with pytest.raises(RuntimeError, match="between 0 and number of dimensions"):
        lsq(x, torch.ones(8), torch.zeros(8), axis=2, is_perchannel=True)
    with pytest.raises(RuntimeError, match="not consistent with input tensor"):
        lsq(x, torch.ones(3), torch.zeros(3), is_perchannel=True)
    # defaults: type range falls back to the quant range (functional.py:92-93)
    y = lsq(x * 100, s, b)
    assert float(y.max()) <= 255.0 and float(y.min()) >= 0.0
    # autograd wiring incl. the size-1 `repeat` path, against the reference goldens
    for case in manifest["cases"]:
        if not case["name"].startswith(("pc_repeat", "pt_affine7", "pc_axis0_sym_float32")):
            continue
        k, p = case["key"], case["params"]
        xs = torch.from_numpy(arrays[k + "x"]).requires_grad_(True)
        sc = torch.from_numpy(arrays[k + "scale"]).requires_grad_(True)
        sh = torch.from_numpy(arrays[k + "shift"]).requires_grad_(True)
        y = lsq(xs, sc, sh, p["quant_min"], p["quant_max"], p["type_min"], p["type_max"], p["axis"], p["use_grad_scaling"],
                p["grad_scaler"], p["is_affine"], p["is_perchannel"], p["eval_mode"], p["init_mode"])
        y.backward(torch.from_numpy(arrays[k + "g"]))
        assert y.detach().numpy().tobytes() == arrays[k + "y"].tobytes(), case["name"]
        assert xs.grad.numpy().tobytes() == arrays[k + "dx"].tobytes(), case["name"]
        assert sc.grad.shape == sc.shape and sh.grad.shape == sh.shape
        np.testing.assert_allclose(sc.grad.numpy(), arrays[k + "ds"], rtol=2e-6, atol=1e-6 * float(arrays[k + "abs_ds"].max()))
        np.testing.assert_allclose(sh.grad.numpy(), arrays[k + "db"], rtol=2e-6, atol=1e-6 * float(arrays[k + "abs_db"].max()) + 1e-30)
    # double backward is refused like the reference (lsq_autograd.cpp:106)
    xs = torch.randn(8, requires_grad=True)
    g = torch.ones(8, requires_grad=True)
    dx, ds, db = torch.ops.torchlsq.lsq_backward_per_tensor(g, xs, s, b, 0, 127, 0, 255, True, 1.0, False, False, False)
    with pytest.raises(RuntimeError, match="double backwards on lsq_per_tensor not supported"):
        dx.sum().backward()


def test_layout_helpers():
    from torchlsq import extension as E
    x = torch.empty(4, 16, 6, 10)
    xd, order = E._dense(x)
    assert xd is x and E._ocl(x, order, 1) == (4, 16, 60) and E._ocl(x, order, 0) == (1, 4, 960)
    cl = x.contiguous(memory_format=torch.channels_last)
    xd, order = E._dense(cl)
    assert xd is cl and E._ocl(cl, order, 1) == (4 * 6 * 10, 16, 1)
    t = x.permute(1, 0, 2, 3)                     # dense, permuted
    xd, order = E._dense(t)
    assert xd is t and E._ocl(t, order, 0) == (4, 16, 60)
    sl = x[:, ::2]                                # not dense -> contiguous copy
    xd, order = E._dense(sl)
    assert xd is not sl and xd.is_contiguous() and E._ocl(xd, order, 1) == (4, 8, 60)
    one = torch.empty(6, 1, 9)
    xd, order = E._dense(one)
    assert E._ocl(one, order, 1) in ((1, 1, 54), (6, 1, 9))      # a single channel either way
    g = torch.empty(4, 16, 6, 10)
    assert E._like_layout(g, cl).stride() == cl.stride()


def test_picklable_factory():
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver
    from torchlsq.quantized import LSQFakeQuantizer
    f = LSQFakeQuantizer.with_args(observer=MovingAverageMinMaxObserver, otype="activation")
    m = f()
    m2 = pickle.loads(pickle.dumps(m))
    assert isinstance(m2, LSQFakeQuantizer) and m2.quant_max == 127


def test_host_mirror_tracks_buffers(oracle_cpu_backend):
    """decisions read a host mirror of the four state buffers; it follows the methods, load_state_dict and train()/eval()"""
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver as Obs
    from torchlsq.quantized import LSQFakeQuantizer as Q
    a = Q(Obs, "activation", init_batches=2)
    x = torch.rand(4, 8) + 0.5
    for _ in range(5):
        a(x)
    mirror = lambda m: (m._h["fake_quant"], m._h["observer"], m._h["learning"], m._h["batch"])
    bufs = lambda m: (int(m.fake_quant_enabled[0]), int(m.observer_enabled[0]), int(m.learning_enabled[0]), int(m.current_batch[0]))
    assert mirror(a) == bufs(a) == (1, 0, 1, 3)
    a.disable_fake_quant(); a.enable_static_estimate()
    assert mirror(a) == bufs(a) == (0, 1, 0, 3)
    b = Q(Obs, "activation", init_batches=2)
    b(x)
    b.load_state_dict(a.state_dict())
    assert mirror(b) == bufs(b) == (0, 1, 0, 3)
    b.fake_quant_enabled[0] = 1          # an out-of-band write is picked up at the next train()/eval() ...
    b.train()
    assert mirror(b) == bufs(b) == (1, 1, 0, 3)
    # ... and at the next forward (the reference reads the buffers on every call): in-place writes, copies from
    # another module and replaced buffers are all seen through the buffers' identity + version counters
    b.fake_quant_enabled.fill_(0)
    y = b(x)
    assert mirror(b)[0] == 0 and y is x                     # fake-quant off: the input passes through
    b.fake_quant_enabled[0] = 1
    assert not torch.equal(b(x), x) and mirror(b)[0] == 1
    b.observer_enabled.copy_(a.fake_quant_enabled)          # a's flag is 0
    b(x)
    assert mirror(b) == bufs(b)
    b.learning_enabled = torch.ones_like(b.learning_enabled)  # buffer replaced (what .to(device) does)
    b(x)
    assert mirror(b) == bufs(b) and mirror(b)[2] == 1
    import copy, pickle
    assert mirror(copy.deepcopy(b)) == mirror(b) and mirror(pickle.loads(pickle.dumps(b))) == mirror(b)


def test_ops_trace_under_torch_compile(oracle_cpu_backend):
    """every op has a shape-only (fake) kernel, so the dispatcher path traces (aot_eager: no codegen involved)"""
    x = torch.randn(4, 8, 6, 6)
    s, b = torch.full((8,), 0.05, requires_grad=True), torch.zeros(8, requires_grad=True)

    def f(x, s, b):
        y = torch.ops.torchlsq.lsq(x, s, b, -8, 7, -128, 127, 1, True, 1.0, True, True, False, False)
        return (y * y).sum()

    ref = f(x, s, b)
    gs_ref = torch.autograd.grad(ref, (s, b))
    out = torch.compile(f, backend="aot_eager")(x, s, b)
    gs = torch.autograd.grad(out, (s, b))
    assert torch.equal(out, ref) and all(torch.equal(a, c) for a, c in zip(gs, gs_ref))
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        fx = torch.empty(4, 8, 6, 6)
        mn, mx = torch.ops.torchlsq.lsq_minmax_per_channel(fx, 1)
        yq, q = torch.ops.torchlsq.lsq_quantize_per_tensor(fx, torch.empty(1), torch.empty(1), 0, 127, 0, 255, 0)
        assert mn.shape == (8,) and q.dtype == torch.int8 and q.shape == fx.shape


def test_lsq_foreach_on_cpu_tensors_is_the_loop(oracle_cpu_backend):
    """CPU tensors are not fused: lsq_foreach is then exactly one lsq call per tensor"""
    from torchlsq.functional import lsq, lsq_foreach
    torch.manual_seed(0)
    xs = [torch.randn(6, 4, 3, 3) * 0.05, torch.randn(5, 8) * 0.05]
    ss = [torch.full((6,), 1e-3), torch.full((5,), 2e-3)]
    bs = [torch.zeros(6), torch.zeros(5)]
    kw = dict(quant_min=-128, quant_max=127, type_min=-128, type_max=127, is_affine=False)
    got = lsq_foreach([x.clone().requires_grad_(True) for x in xs], ss, bs, axis=0, **kw)
    want = [lsq(x, s, b, axis=0, is_perchannel=True, **kw) for x, s, b in zip(xs, ss, bs)]
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_single_launch_policy_modes():
    """TORCHLSQ_SINGLE_LAUNCH_BACKWARD / set_single_launch_backward: "auto" (default) = per-tensor tensors of at most 8 MB,
    True = always, False = never (host logic only: no GPU needed)"""
    from torchlsq import extension as E
    saved = E._SINGLE_LAUNCH_BWD[0]
    try:
        E.set_single_launch_backward("auto")
        assert E._wants_ticket(1) and E._wants_ticket(8 << 20) and not E._wants_ticket((8 << 20) + 1)
        E.set_single_launch_backward(True)
        assert E._wants_ticket(1 << 40)
        E.set_single_launch_backward(False)
        assert not E._wants_ticket(1)
    finally:
        E._SINGLE_LAUNCH_BWD[0] = saved
