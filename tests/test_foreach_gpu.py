"""GPU: many per-channel quantizers in one launch (lsq_hip_*_per_channel_multi -> functional.lsq_foreach -> LSQWeightGroup)
against the same tensors through single calls: outputs and every gradient bit-identical (one exception, parameter gradients of
16-bit short rows on very many channels: test_short_rows_on_many_channels_fuse_although_single_calls_take_windows)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bits(t):
    t = t.detach().contiguous()
    return t.view({1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[t.element_size()]).cpu().numpy().tobytes()


def _weights(shapes, dtype, dev, seed):
    from torchlsq import synth
    xs, gs, ss, bs = [], [], [], []
    for k, shape in enumerate(shapes):
        n = int(np.prod(shape))
        xs.append(synth.normal_like(n, seed + 4 * k, 0.0, 0.05, dtype=dtype, device=dev).view(shape))
        gs.append(synth.normal_like(n, seed + 4 * k + 1, 0.0, 1e-3, dtype=dtype, device=dev).view(shape))
        pdt = torch.float64 if dtype == torch.float64 else torch.float32
        ss.append(synth.uniform_like(shape[0], seed + 4 * k + 2, 5e-4, 2.5e-3, device=dev, dtype=pdt))
        bs.append(synth.normal_like(shape[0], seed + 4 * k + 3, 0.0, 1e-3, device=dev, dtype=pdt))
    return xs, gs, ss, bs


def _run(fn_each, xs, gs, ss, bs, kw, fused):
    from torchlsq.functional import lsq, lsq_foreach
    xl = [x.clone().requires_grad_(True) for x in xs]
    sl = [s.clone().requires_grad_(True) for s in ss]
    bl = [b.clone().requires_grad_(True) for b in bs]
    if fused:
        ys = lsq_foreach(xl, sl, bl, axis=0, **kw)
    else:
        ys = [lsq(x, s, b, axis=0, is_perchannel=True, **kw) for x, s, b in zip(xl, sl, bl)]
    torch.autograd.backward(ys, gs)
    torch.cuda.synchronize()
    return ys, [x.grad for x in xl], [s.grad for s in sl], [b.grad for b in bl]


@pytest.mark.parametrize("mode", ["sym", "affine", "eval", "init"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float64, torch.float16])
def test_foreach_equals_single_calls_bit_for_bit(dtype, mode):
    """conv / linear weight shapes of a CNN and a transformer block, a few that the multi-tensor kernels do not take
    ([64,3,7,7]: 147 elements per channel, not packet-aligned; [10,64]: one short row per channel) and more than one launch's
    worth (32) of them"""
    from torchlsq import extension as E
    dev = torch.device("cuda:0")
    shapes = [(512, 512, 3, 3), (256, 128, 3, 3), (64, 3, 7, 7), (1024, 4096), (4096, 1024), (128, 64, 1, 1), (10, 64),
              (64, 64, 3, 3)] + [(256, 256, 3, 3)] * 30 + [(2048, 2048)]
    xs, gs, ss, bs = _weights(shapes, dtype, dev, 4000)
    taken = [E.hip_multi_eligible(x, 0) for x in xs]
    assert sum(taken) >= 33 and not all(taken), taken       # both routes, and more than one launch of 32
    kw = dict(quant_min=-128, quant_max=127, type_min=-128, type_max=127, is_affine=(mode != "sym"),
              eval_mode=(mode == "eval"), init_mode=(mode == "init"), use_grad_scaling=True, grad_scaler=0.5)
    single = _run(None, xs, gs, ss, bs, kw, fused=False)
    fused = _run(None, xs, gs, ss, bs, kw, fused=True)
    for what, a, b in zip(("y", "dx", "d_scale", "d_shift"), single, fused):
        for i, (u, v) in enumerate(zip(a, b)):
            if u is None or v is None:
                assert u is None and v is None, (what, i)
                continue
            assert u.shape == v.shape and _bits(u) == _bits(v), "%s of tensor %d %s differs (%s, %s)" % (what, i, shapes[i], dtype, mode)


def test_short_rows_on_many_channels_fuse_although_single_calls_take_windows():
    """16-bit weights with rows under half a workgroup's span on more than 8 x CUs channels ([3072,768], [4096,576]): a single
    call takes the window kernels (the walk is 11-32 % behind there, profiles/r04_seg_weights.txt), the multi-tensor launch
    still takes them as one-workgroup-per-channel segments (one launch is worth more): y and dx are the single calls' bits,
    d_scale / d_shift are summed in another order -- both within the parity bar of the oracle"""
    from helpers import assert_reduction_close
    from oracle import lsq_oracle as O
    from torchlsq import extension as E
    dev = torch.device("cuda:0")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    shapes = [(12 * cus, 768), (16 * cus, 576), (9 * cus, 768), (512, 512, 3, 3)]
    dtype = torch.bfloat16
    xs, gs, ss, bs = _weights(shapes, dtype, dev, 5100)
    assert all(E.hip_multi_eligible(x, 0) for x in xs)
    kw = dict(quant_min=-128, quant_max=127, type_min=-128, type_max=127, is_affine=True, use_grad_scaling=True, grad_scaler=1.0)
    single = _run(None, xs, gs, ss, bs, kw, fused=False)
    fused = _run(None, xs, gs, ss, bs, kw, fused=True)
    for i, shape in enumerate(shapes):
        assert _bits(single[0][i]) == _bits(fused[0][i]) and _bits(single[1][i]) == _bits(fused[1][i]), shape
        outer, C, inner = O.axis_to_ocl(shape, 0)
        r = O.bwd_pc(gs[i].float().cpu().numpy(), xs[i].float().cpu().numpy(), ss[i].cpu().numpy(), bs[i].cpu().numpy(), outer, C, inner,
                     -128, 127, -128, 127, True, 1.0, False, False, False)
        for route in (single, fused):
            assert_reduction_close(route[2][i].cpu().numpy(), r.ds_wide, r.abs_ds, "%s ds" % (shape,))
            assert_reduction_close(route[3][i].cpu().numpy(), r.db_wide, r.abs_db, "%s db" % (shape,))


def test_fifty_conv_weights():
    """the review's case: 50 x [512,512,3,3] fp32 qint8 per-channel weights (BASELINE config 3), forward + backward"""
    dev = torch.device("cuda:0")
    shapes = [(512, 512, 3, 3)] * 50
    xs, gs, ss, bs = _weights(shapes, torch.float32, dev, 9000)
    kw = dict(quant_min=-128, quant_max=127, type_min=-128, type_max=127, is_affine=False)
    single = _run(None, xs, gs, ss, bs, kw, fused=False)
    fused = _run(None, xs, gs, ss, bs, kw, fused=True)
    for a, b in zip(single[:3], fused[:3]):
        for u, v in zip(a, b):
            assert _bits(u) == _bits(v)
    # ... and against the reference digest of config 3 for a tensor generated like the golden one
    from torchlsq import synth
    import json, os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_digests.json")))["configs"]["cfg3"]
    x, g, scale, shift = synth.make_inputs("cfg3", device=dev, dtype=torch.float32)
    from helpers import sha

    def sha_t(t):
        return sha(t.detach().contiguous().cpu().numpy())
    ys, dxs, dss, _ = _run(None, [x, x.clone()], [g, g.clone()], [scale, scale.clone()], [shift, shift.clone()],
                            dict(quant_min=-128, quant_max=127, type_min=-128, type_max=127, is_affine=False), fused=True)
    for y, dx in zip(ys, dxs):
        assert sha_t(y) == d["y_sha256"] and sha_t(dx) == d["dx_sha256"]


def test_weight_group_on_a_qat_model():
    """prepare_qat model with LSQFakeQuantizer weight quantizers: with the LSQWeightGroup hook the layers get their weights from
    one fused call; loss, input gradient and every parameter gradient equal the ungrouped model's bit for bit"""
    import copy
    import torchlsq  # noqa: F401
    from torch.ao.quantization import QConfig, prepare_qat
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torchlsq.quantized import LSQFakeQuantizer, LSQWeightGroup
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    net = torch.nn.Sequential(
        torch.nn.Conv2d(8, 256, 3, padding=1), torch.nn.ReLU(),
        torch.nn.Conv2d(256, 256, 3, padding=1), torch.nn.ReLU(),
        torch.nn.Conv2d(256, 512, 3, padding=1, stride=2), torch.nn.ReLU(),
        torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(512, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 16))
    net.qconfig = QConfig(
        activation=LSQFakeQuantizer.with_args(observer=MovingAverageMinMaxObserver, otype="activation", init_batches=1),
        weight=LSQFakeQuantizer.with_args(observer=MovingAveragePerChannelMinMaxObserver, otype="weight", dtype=torch.qint8,
                                          qscheme=torch.per_channel_symmetric))
    net = prepare_qat(net.train()).to(dev)
    x0 = torch.randn(4, 8, 16, 16, device=dev)
    for _ in range(4):                       # creating call + initialisation batches: every quantizer reaches its steady state
        net(x0).sum().backward()
    net.zero_grad(set_to_none=True)
    grouped = copy.deepcopy(net)
    group = LSQWeightGroup(grouped)
    assert len(group.pairs) == 5
    x = torch.randn(4, 8, 16, 16, device=dev)
    outs = []
    for model in (net, grouped):
        seen = {}
        hooks = [m.weight_fake_quant.register_forward_hook(lambda mod, a, out, k=k: seen.__setitem__(k, out.detach().clone()))
                 for k, m in enumerate(mm for mm in model.modules() if hasattr(mm, "weight_fake_quant"))]
        xi = x.clone().requires_grad_(True)
        loss = (model(xi) ** 2).sum()
        loss.backward()
        torch.cuda.synchronize()
        for h in hooks:
            h.remove()
        outs.append((loss, xi.grad, {n: p.grad for n, p in model.named_parameters()}, seen))
    assert group.last_fused >= 3, group.last_fused        # the weights of the big layers went through the fused call
    # forward: every layer got the same fake-quantized weight, the loss is the same number
    assert sorted(outs[0][3]) == sorted(outs[1][3]) and len(outs[0][3]) == 5
    for k in outs[0][3]:
        assert _bits(outs[0][3][k]) == _bits(outs[1][3][k]), k
    assert _bits(outs[0][0]) == _bits(outs[1][0])
    # backward: the gradients that reach the quantizers come out of the framework's convolution / GEMM backward kernels, which
    # are not bit-reproducible run to run; the fused node itself is (test_foreach_equals_single_calls_bit_for_bit), so here:
    # same gradients to rounding noise, for every parameter (scale / shift of all quantizers included) and the input
    def close(a, b, what):
        assert (a is None) == (b is None), what
        if a is not None:
            tol = 1e-4 * float(a.abs().max()) + 1e-12
            assert float((a - b).abs().max()) <= tol, (what, float((a - b).abs().max()), tol)
    close(outs[0][1], outs[1][1], "input gradient")
    for n, g0 in outs[0][2].items():
        close(g0, outs[1][2][n], n)


@pytest.mark.parametrize("seed", range(6))
def test_foreach_random_weight_lists(seed):
    """random lists of weight-like tensors (channels first, any row length: whole spans, short rows the segment walk takes,
    rows it leaves to the window kernels, rows that are not packet-aligned), more than one launch's worth, random modes:
    lsq_foreach == the single calls, bit for bit, on both host layers"""
    from torchlsq import extension as E
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7000 + seed)
    dtype = [torch.float32, torch.bfloat16, torch.float16][seed % 3]      # (fp64 outputs of the window kernels agree run to run
                                                                           # only to fp64 rounding: DESIGN.md, Reproducibility)
    n_tensors = int(rng.integers(2, 70))
    shapes = []
    for _ in range(n_tensors):
        C = int(rng.choice([1, 3, 8, 64, 100, 256, 500]))
        tail = [int(rng.choice([1, 2, 3, 7, 8, 16, 64, 96, 128, 576, 1000, 1024, 2048, 4608]))]
        if rng.random() < 0.4:
            tail.append(int(rng.choice([1, 3, 4])))
        if C * int(np.prod(tail)) > 1_500_000:
            tail = [128]
        shapes.append((C, *tail))
    xs, gs, ss, bs = _weights(shapes, dtype, dev, 100 * seed)
    mode = ["sym", "affine", "eval", "init"][int(rng.integers(0, 4))]
    kw = dict(quant_min=-128, quant_max=127, type_min=-128, type_max=127, is_affine=(mode != "sym"),
              eval_mode=(mode == "eval"), init_mode=(mode == "init"), use_grad_scaling=bool(rng.random() < 0.8),
              grad_scaler=float(rng.choice([1.0, 0.5])))
    saved = E.host_binding()
    try:
        for binding in (("native", "ctypes") if E.native_lsq() is not None or saved == "native" else ("ctypes",)):
            E.set_host_binding(binding)
            single = _run(None, xs, gs, ss, bs, kw, fused=False)
            fused = _run(None, xs, gs, ss, bs, kw, fused=True)
            for what, a, b in zip(("y", "dx", "d_scale", "d_shift"), single, fused):
                for i, (u, v) in enumerate(zip(a, b)):
                    if u is None or v is None:
                        assert u is None and v is None, (what, i)
                        continue
                    assert u.shape == v.shape and _bits(u) == _bits(v), "%s of tensor %d %s differs (%s, %s, %s)" % (
                        what, i, shapes[i], dtype, mode, binding)
    finally:
        E.set_host_binding(saved)


@pytest.mark.parametrize("binding", ["native", "ctypes"])
@pytest.mark.parametrize("init_mode", [False, True])
def test_unused_outputs_of_a_fused_call_get_no_gradients(binding, init_mode):
    """an output nobody used has no upstream gradient: its tensor takes no part in the backward launch and gets NO gradients,
    exactly what N separate lsq calls give it -- with init_mode the parameter gradients ignore the upstream gradient
    (lsq_kernel.h:116), so a zero-filled stand-in would have invented d_scale / d_shift for it"""
    from torchlsq import extension as E
    from torchlsq.functional import lsq, lsq_foreach
    if binding == "native" and E.native_lsq() is None:
        pytest.skip("the C++ binding is not built")
    dev = torch.device("cuda:0")
    saved = E.host_binding()
    E.set_host_binding(binding)
    try:
        shapes = [(64, 64, 3, 3), (128, 64, 3, 3), (96, 256)]
        xs, gs, ss, bs = _weights(shapes, torch.float32, dev, 77)
        kw = dict(quant_min=-128, quant_max=127, type_min=-128, type_max=127, is_affine=True, init_mode=init_mode)

        def run(fused):
            xl = [x.clone().requires_grad_(True) for x in xs]
            sl = [s.clone().requires_grad_(True) for s in ss]
            bl = [b.clone().requires_grad_(True) for b in bs]
            ys = lsq_foreach(xl, sl, bl, axis=0, **kw) if fused else [lsq(x, s, b, axis=0, is_perchannel=True, **kw) for x, s, b in zip(xl, sl, bl)]
            torch.autograd.backward([ys[0], ys[2]], [gs[0], gs[2]])         # output 1 is never used
            torch.cuda.synchronize()
            return xl, sl, bl
        fx, fs, fb = run(True)
        ex, es, eb = run(False)
        assert fx[1].grad is None and fs[1].grad is None and fb[1].grad is None
        assert ex[1].grad is None and es[1].grad is None
        for i in (0, 2):
            assert _bits(fx[i].grad) == _bits(ex[i].grad) and _bits(fs[i].grad) == _bits(es[i].grad) and _bits(fb[i].grad) == _bits(eb[i].grad)
    finally:
        E.set_host_binding(saved)


def test_foreach_with_cpu_entries_takes_the_python_partition():
    """a list with tensors in host memory: those go through `lsq` (liblsq_cpu.so), the GPU ones still fuse -- on either host layer"""
    from torchlsq.functional import lsq, lsq_foreach
    dev = torch.device("cuda:0")
    xs, gs, ss, bs = _weights([(64, 64, 3, 3), (128, 64, 3, 3), (32, 16, 3, 3)], torch.float32, dev, 91)
    xs[2], ss[2], bs[2] = xs[2].cpu(), ss[2].cpu(), bs[2].cpu()
    kw = dict(quant_min=-128, quant_max=127, type_min=-128, type_max=127, is_affine=False)
    ys = lsq_foreach(xs, ss, bs, axis=0, **kw)
    assert ys[2].device.type == "cpu" and ys[0].is_cuda
    for x, s, b, y in zip(xs, ss, bs, ys):
        assert _bits(y) == _bits(lsq(x, s, b, axis=0, is_perchannel=True, **kw))


def test_weight_group_results_are_dropped_when_stale_or_unused():
    """LSQWeightGroup's stash is valid for exactly the weight and parameter VALUES it was computed from (Tensor._version): an
    in-place update between prequantize() and the layer's call recomputes; whatever the layers did not pick up is dropped when
    the model's forward ends, so the model stays deep-copyable and picklable"""
    import copy
    from torch.ao.quantization import QConfig
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torchlsq.quantized import LSQFakeQuantizer, LSQWeightGroup
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Conv2d(16, 64, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(64, 64, 3, padding=1))
    model.qconfig = QConfig(activation=LSQFakeQuantizer.with_args(observer=MovingAverageMinMaxObserver, otype="activation", init_batches=1),
                            weight=LSQFakeQuantizer.with_args(observer=MovingAveragePerChannelMinMaxObserver, otype="weight", dtype=torch.qint8,
                                                              qscheme=torch.per_channel_symmetric))
    torch.ao.quantization.prepare_qat(model.train(), inplace=True)
    model.to(dev)
    x = torch.randn(4, 16, 12, 12, device=dev)
    for _ in range(3):
        model(x)
    group = LSQWeightGroup(model, register_hook=False)
    assert group.prequantize() == 2
    wq = model[0].weight_fake_quant
    stale = wq._prefetched[-1].detach().clone()
    with torch.no_grad():
        model[0].weight.mul_(1.5)                       # what optimizer.step() does: in place, bumps the version
    fresh = wq(model[0].weight)
    assert wq._prefetched is None and not torch.equal(fresh, stale)
    plain = LSQFakeQuantizer.forward(wq, model[0].weight)
    assert torch.equal(fresh, plain)
    # with the hooks: a forward that leaves a result unused (here: we never call layer 2) must not leave it behind
    group2 = LSQWeightGroup(model)
    group2.prequantize()
    assert model[2].weight_fake_quant._prefetched is not None
    model(x)                                            # pre-hook stashes, layers consume, post-hook clears
    assert all(q._prefetched is None for _, q in group2.pairs)
    group2.prequantize()
    group2.remove()
    assert all(q._prefetched is None for _, q in group2.pairs)
    copy.deepcopy(model)                                # no stashed non-leaf tensor in the way
