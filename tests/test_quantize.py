"""Real quantized tensors from a trained LSQ quantizer: `torchlsq.functional.lsq_quantize`, `LSQFakeQuantizer.quantize`
and the levels-only forward behind them (`torch.ops.torchlsq.lsq_levels_*`: lsq_hip_forward_* with y == NULL).

What is held: the byte stream is the digest-pinned integer levels of the forward (tests/golden/config_digests.json,
generated from the reference), `q.dequantize()` is the fake-quantized output bit for bit, and the quantizer constants are
the kernels' (lsq_cpu.cpp:44-47, lsq_kernel.h:12,157-158).  CPU tensors take the torch-ops formula of _cpu_host.cpu_levels,
held here to the product's CPU forward; GPU tensors the HIP kernels.
"""
import numpy as np
import pytest
import torch

from helpers import sha


def _cases():
    g = torch.Generator().manual_seed(11)
    x = torch.randn(6, 8, 5, 7, generator=g) * 1.5 + 0.4
    x.view(-1)[:6] = torch.tensor([float("nan"), float("inf"), float("-inf"), 0.0, -0.0, 1e-30])
    pt = (torch.tensor([0.031]), torch.tensor([0.27]))
    pc = (torch.rand(8, generator=g) * 0.2 + 0.01, torch.randn(8, generator=g) * 0.3)
    pc[0][3] = -pc[0][3]            # a negative scale: the kernels use |scale|
    return x, pt, pc


@pytest.mark.parametrize("dtype,qr", [(torch.quint8, (0, 255)), (torch.quint8, (0, 127)), (torch.qint8, (-128, 127)), (torch.qint8, (-8, 7))])
def test_cpu_quantize_dequantizes_to_the_fake_quant_output(oracle_cpu_backend, dtype, qr):
    from torchlsq.functional import lsq, lsq_quantize
    x, pt, pc = _cases()
    tmin, tmax = (0, 255) if dtype == torch.quint8 else (-128, 127)
    for (s, b), kw in ((pt, dict()), (pc, dict(axis=1, is_perchannel=True))):
        y = lsq(x, s, b, qr[0], qr[1], tmin, tmax, **kw)
        q = lsq_quantize(x, s, b, qr[0], qr[1], tmin, tmax, dtype=dtype, **kw)
        assert q.dtype == dtype and q.is_quantized and q.shape == x.shape
        assert y.numpy().tobytes() == q.dequantize().numpy().tobytes()
        assert int(q.int_repr().min()) >= qr[0] and int(q.int_repr().max()) <= qr[1]


def test_cpu_module_quantize(oracle_cpu_backend):
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver as Obs, MovingAveragePerChannelMinMaxObserver as PObs
    from torchlsq.quantized import LSQFakeQuantizer as Q
    x, _, _ = _cases()
    x = x.nan_to_num(0.0, 3.0, -3.0)
    a = Q(Obs, "activation", init_batches=1)
    for _ in range(4):
        y = a(x)
    q = a.quantize(x)
    assert q.dtype == torch.quint8 and torch.equal(q.dequantize(), y.detach())
    w = Q(PObs, "weight", dtype=torch.qint8, qscheme=torch.per_channel_symmetric)
    wt = torch.randn(8, 4, 3, 3) * 0.1
    w(wt)
    yw = w(wt)
    qw = w.quantize(wt)
    assert qw.dtype == torch.qint8 and qw.qscheme() == torch.per_channel_affine and qw.q_per_channel_axis() == 0
    assert torch.equal(qw.dequantize(), yw.detach())
    with pytest.raises(AssertionError, match="at least one batch"):
        Q(Obs, "activation").quantize(x)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfg1", "cfg3", "cfg5_fp32", "cfg2"])
def test_levels_only_stream_is_the_digest_pinned_levels(config_digests, name):
    """lsq_hip_forward_* with y == NULL: the int8 stream alone, bit-identical to the reference-derived digest at BASELINE
    sizes; level_bias 0 gives the uint8 int_repr for quint8 ranges beyond 127."""
    from torchlsq import synth
    dev = torch.device("cuda:0")
    d = config_digests[name]
    p = d["params"]
    x, _, scale, shift = synth.make_inputs(d["config"], device=dev, dtype=torch.float32)
    args = (p["quant_min"], p["quant_max"], p["type_min"], p["type_max"])
    for bias in sorted({0, 128 if p["quant_max"] > 127 else 0}):
        if p["is_perchannel"]:
            q = torch.ops.torchlsq.lsq_levels_per_channel(x, scale, shift, p["axis"], *args, bias)
        else:
            q = torch.ops.torchlsq.lsq_levels_per_tensor(x, scale, shift, *args, bias)
        assert q.dtype == torch.int8 and q.shape == x.shape
        lv = (q.view(torch.uint8).to(torch.int16) if (bias == 0 and p["quant_min"] >= 0) else q.to(torch.int16)) + bias
        assert sha(lv.cpu().numpy()) == d["levels_int16_sha256"], (name, bias)


@pytest.mark.gpu
@pytest.mark.parametrize("io", [torch.float32, torch.bfloat16, torch.float16, torch.float64])
def test_gpu_quantize_dequantizes_to_the_fake_quant_output(io):
    from torchlsq.functional import lsq, lsq_quantize
    dev = torch.device("cuda:0")
    x, pt, pc = _cases()
    pd = torch.float64 if io == torch.float64 else torch.float32
    shapes = [x, torch.randn(3, 8, 1031) * 2, torch.randn(64, 8, 56, 56), torch.randn(1000, 8)]
    for xx in shapes:
        xx = xx.to(dev).to(io)
        for dtype, qr in ((torch.quint8, (0, 255)), (torch.qint8, (-8, 7))):
            tmin, tmax = (0, 255) if dtype == torch.quint8 else (-128, 127)
            for (s, b), kw in ((pt, dict()), (pc, dict(axis=1, is_perchannel=True))):
                s, b = s.to(dev).to(pd), b.to(dev).to(pd)
                y = lsq(xx, s, b, qr[0], qr[1], tmin, tmax, **kw)
                q = lsq_quantize(xx, s, b, qr[0], qr[1], tmin, tmax, dtype=dtype, **kw)
                assert q.dtype == dtype and q.is_cuda
                deq = q.dequantize()
                if io == torch.float64:     # a quantized tensor dequantizes in fp32: the levels are exact (below), the values close
                    assert torch.allclose(deq.double(), y, rtol=1e-6, atol=0), (io, dtype, kw, tuple(xx.shape))
                else:                       # fp32 bit for bit; 16-bit storage: y is that fp32 value rounded to the storage type
                    assert torch.equal(deq.to(io), y), (io, dtype, kw, tuple(xx.shape))
                # the levels equal those of the y-writing forward
                if kw:
                    _, q2 = torch.ops.torchlsq.lsq_quantize_per_channel(xx, s, b, 1, qr[0], qr[1], tmin, tmax, 128 if qr[1] > 127 else 0)
                else:
                    _, q2 = torch.ops.torchlsq.lsq_quantize_per_tensor(xx, s, b, qr[0], qr[1], tmin, tmax, 128 if qr[1] > 127 else 0)
                want = q2.to(torch.int16) + (128 if qr[1] > 127 else 0)
                assert torch.equal(q.int_repr().to(torch.int16), want)


@pytest.mark.gpu
def test_gpu_levels_only_layouts_and_unaligned():
    """channels-last, transposed, sliced (unaligned) inputs through the levels-only forward == the y-writing forward's levels"""
    dev = torch.device("cuda:0")
    base = torch.randn(8, 16, 9, 11, device=dev)
    s = torch.rand(16, device=dev) * 0.1 + 0.01
    b = torch.randn(16, device=dev) * 0.05
    views = [base.contiguous(memory_format=torch.channels_last), base.transpose(0, 1), base.reshape(-1)[3:3 + 16 * 700].view(16, 700).t()]
    for v in views:
        axis = 1 if v.dim() == 4 and v.shape[1] == 16 else (0 if v.shape[0] == 16 else 1)
        _, want = torch.ops.torchlsq.lsq_quantize_per_channel(v, s, b, axis, -8, 7, -128, 127, 0)
        got = torch.ops.torchlsq.lsq_levels_per_channel(v, s, b, axis, -8, 7, -128, 127, 0)
        assert torch.equal(got, want) and got.stride() == want.stride()
        _, want = torch.ops.torchlsq.lsq_quantize_per_tensor(v, s[:1], b[:1], 0, 255, 0, 255, 128)
        got = torch.ops.torchlsq.lsq_levels_per_tensor(v, s[:1], b[:1], 0, 255, 0, 255, 128)
        assert torch.equal(got, want)


def test_cpu_torch_convert_after_qat_reproduces_the_fake_quantized_model(oracle_cpu_backend):
    """The step after the path: a QAT model whose QConfig holds LSQFakeQuantizers goes through torch's own
    `torch.ao.quantization.convert` -- which reads `calculate_qparams()`, `qscheme`, `ch_axis`, `dtype` of every quantizer
    (reference quantized/modules/observers.py:378-422) -- and the resulting int8 model (quantized conv kernels of the CPU
    backend) reproduces the fake-quantized model's output to within one output level (here: exactly)."""
    import warnings
    from torch.ao.quantization import DeQuantStub, QConfig, QuantStub, convert, prepare_qat
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torchlsq.quantized import LSQFakeQuantizer

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.dq = QuantStub(), DeQuantStub()
            self.c1 = torch.nn.Conv2d(3, 8, 3, padding=1)
            self.r = torch.nn.ReLU()
            self.c2 = torch.nn.Conv2d(8, 4, 3, padding=1)

        def forward(self, x):
            return self.dq(self.c2(self.r(self.c1(self.q(x)))))

    torch.manual_seed(0)
    m = Net()
    m.qconfig = QConfig(activation=LSQFakeQuantizer.with_args(observer=MovingAverageMinMaxObserver, otype="activation", init_batches=3),
                        weight=LSQFakeQuantizer.with_args(observer=MovingAveragePerChannelMinMaxObserver, otype="weight", dtype=torch.qint8,
                                                          qscheme=torch.per_channel_symmetric))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        prepare_qat(m.train(), inplace=True)
        x = torch.randn(8, 3, 16, 16)
        opt = None
        for i in range(8):
            y = m(x)
            if i == 0:                              # the quantizers' parameters exist after the first call
                opt = torch.optim.SGD(m.parameters(), lr=1e-3)
            opt.zero_grad()
            (y ** 2).mean().backward()
            opt.step()
        m.eval()
        y_fake = m(x)
        mq = convert(m, inplace=False)
        y_int8 = mq(x)
    assert type(mq.c1).__name__ == "Conv2d" and "quantized" in type(mq.c1).__module__
    out_scale = float(mq.c2.scale)
    assert float((y_fake - y_int8).abs().max()) <= out_scale * 1.001, (float((y_fake - y_int8).abs().max()), out_scale)
    # the weights torch quantized with the module's qparams are the module's own levels
    wq = m.c1.weight_fake_quant
    assert torch.equal(mq.c1.weight().int_repr(), wq.quantize(m.c1.weight).int_repr())


def test_cpu_fx_graph_mode_qat_and_convert(oracle_cpu_backend):
    """the same through FX graph mode (`prepare_qat_fx` inserts the quantizers from the QConfigMapping, `convert_fx` lowers to
    the int8 kernels): the module is a drop-in `activation_post_process` / `weight_fake_quant` there too"""
    import warnings
    from torch.ao.quantization import QConfig, QConfigMapping
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torch.ao.quantization.quantize_fx import convert_fx, prepare_qat_fx
    from torchlsq.quantized import LSQFakeQuantizer
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 3, padding=1))
    qc = QConfig(activation=LSQFakeQuantizer.with_args(observer=MovingAverageMinMaxObserver, otype="activation", init_batches=3),
                 weight=LSQFakeQuantizer.with_args(observer=MovingAveragePerChannelMinMaxObserver, otype="weight", dtype=torch.qint8,
                                                   qscheme=torch.per_channel_symmetric))
    x = torch.randn(8, 3, 16, 16)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p = prepare_qat_fx(m.train(), QConfigMapping().set_global(qc), (x,))
        assert sum(isinstance(mod, LSQFakeQuantizer) for mod in p.modules()) >= 4
        for _ in range(6):
            p(x)
        p.eval()
        y_fake = p(x)
        y_int8 = convert_fx(p)(x)
    assert float((y_fake - y_int8).abs().max()) <= 0.05 * float(y_fake.abs().max()) + 1e-6
