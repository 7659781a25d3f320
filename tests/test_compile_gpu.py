"""GPU: torch.compile (inductor) over the HIP ops -- SURVEY.md section 8(f)4.

The LSQ ops are opaque custom operators with shape-only (fake) kernels and registered autograd, so inductor schedules
them as extern calls between its own generated kernels.  Compiled forward + backward must equal eager BIT FOR BIT (same
C-ABI calls underneath): the functional entry point per-tensor and per-channel, fp32 and bf16, and LSQFakeQuantizer in
its steady state inside a small compiled block.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _ops():
    import torchlsq  # noqa: F401
    from torchlsq import extension
    extension._assert_has_ops()
    torch._dynamo.reset()
    yield
    torch._dynamo.reset()


def _grads(fn, tensors):
    out = fn(*tensors)
    gs = torch.autograd.grad(out, [t for t in tensors if t.requires_grad])
    return out, gs


@pytest.mark.parametrize("per_channel", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_inductor_compiled_lsq_equals_eager(per_channel, dtype):
    from torchlsq.functional import lsq
    from torchlsq import synth
    dev = torch.device("cuda:0")
    shape = (16, 64, 14, 14)
    n = 16 * 64 * 14 * 14
    x = synth.normal_like(n, 3, 0.4, 1.0, dtype=dtype, device=dev).view(shape).requires_grad_(True)
    if per_channel:
        s = synth.uniform_like(64, 5, 0.02, 0.2, device=dev).requires_grad_(True)
        b = synth.normal_like(64, 6, 0.0, 0.1, device=dev).requires_grad_(True)
        kw = dict(quant_min=-8, quant_max=7, type_min=-128, type_max=127, axis=1, is_perchannel=True)
    else:
        s = torch.tensor([0.03], device=dev, requires_grad=True)
        b = torch.tensor([0.05], device=dev, requires_grad=True)
        kw = dict(quant_min=0, quant_max=127, type_min=0, type_max=255)

    cot = synth.normal_like(n, 4, 0.0, 1e-3, dtype=dtype, device=dev).view(shape)

    def f(x, s, b):
        h = x * 2.0                        # work for inductor's own kernels on both sides of the custom op; scaling by
        y = lsq(h, s, b, **kw)             # powers of two is exact, so eager and compiled code must agree bit for bit
        return y * 0.5

    def run(fn):
        out = fn(x, s, b)
        return out, torch.autograd.grad(out, (x, s, b), grad_outputs=cot)

    ref, g_ref = run(f)
    cf = torch.compile(f, backend="inductor", fullgraph=True)
    for _ in range(2):                     # second call: the cached graph
        out, g = run(cf)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        for a, c in zip(g, g_ref):
            assert a.dtype == c.dtype and torch.equal(a, c)


def test_inductor_compiled_op_level_calls_equal_eager():
    """the four backend ops called directly (what a compiled autograd graph contains)"""
    from torchlsq import synth
    dev = torch.device("cuda:0")
    n = 8 * 32 * 49
    x = synth.normal_like(n, 7, 0.0, 1.0, device=dev).view(8, 32, 7, 7)
    g = synth.normal_like(n, 8, 0.0, 1e-3, device=dev).view(8, 32, 7, 7)
    s1, b1 = torch.tensor([0.05], device=dev), torch.tensor([0.0], device=dev)
    sc, bc = synth.uniform_like(32, 9, 0.02, 0.2, device=dev), synth.normal_like(32, 10, 0.0, 0.1, device=dev)
    ops = torch.ops.torchlsq
    tail = (-8, 7, -128, 127, True, 1.0, False, False, False)

    def f(x, g):
        y1 = ops.lsq_forward_per_tensor(x, s1, b1, *tail)
        d1 = ops.lsq_backward_per_tensor(g, x, s1, b1, *tail)
        y2 = ops.lsq_forward_per_channel(x, sc, bc, 1, *tail)
        d2 = ops.lsq_backward_per_channel(g, x, sc, bc, 1, *tail)
        return (y1 + y2, d1[0] + d2[0], d1[1], d1[2], d2[1], d2[2])

    ref = f(x, g)
    out = torch.compile(f, backend="inductor", fullgraph=True)(x, g)
    torch.cuda.synchronize()
    for a, c in zip(out, ref):
        assert torch.equal(a, c)


def test_inductor_compiled_block_with_fake_quantizers_in_steady_state():
    """LSQFakeQuantizer modules (activation per-tensor + weight per-channel) after their initialisation phase, inside a
    compiled conv block: outputs and all gradients equal eager."""
    from torchlsq.quantized import LSQFakeQuantizer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(8, 16, 3, padding=1)
            self.aq = LSQFakeQuantizer(None, "activation", init_mode="learnable", init_batches=1)
            self.wq = LSQFakeQuantizer(None, "weight", dtype=torch.qint8, qscheme=torch.per_channel_symmetric,
                                       init_mode="learnable")

        def forward(self, x):
            return torch.nn.functional.conv2d(self.aq(x), self.wq(self.conv.weight), self.conv.bias, padding=1)

    m = Block().to(dev)
    x = torch.randn(4, 8, 16, 16, device=dev)
    for _ in range(4):                      # run every quantizer through its initialisation phase, eagerly
        m(x).sum().backward()
    m.zero_grad(set_to_none=True)

    def run(mod):
        xin = x.clone().requires_grad_(True)
        out = mod(xin)
        out.square().sum().backward()
        grads = [xin.grad.clone()] + [p.grad.clone() for p in m.parameters() if p.grad is not None]
        m.zero_grad(set_to_none=True)
        return out.detach(), grads

    ref, g_ref = run(m)
    cm = torch.compile(m, backend="inductor")
    out, g = run(cm)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert len(g) == len(g_ref) and len(g) >= 5        # x, conv weight + bias, activation scale/shift, weight scale
    for a, c in zip(g, g_ref):
        assert torch.allclose(a, c, rtol=1e-5, atol=1e-6)   # conv backward algorithms may differ between graphs


def test_inductor_compiled_levels_only_op_equals_eager():
    """the conversion-time op (int8 levels alone, y == NULL underneath) as an extern call inside a compiled graph"""
    from torchlsq import synth
    dev = torch.device("cuda:0")
    x = synth.normal_like(8 * 32 * 14 * 14, 9, 0.4, 1.0, device=dev).view(8, 32, 14, 14)
    s, b = synth.uniform_like(32, 5, 0.02, 0.2, device=dev), synth.normal_like(32, 6, 0.0, 0.1, device=dev)

    def f(x, s, b):
        h = x * 2.0
        q_pt = torch.ops.torchlsq.lsq_levels_per_tensor(h, s[:1], b[:1], 0, 255, 0, 255, 0)
        q_pc = torch.ops.torchlsq.lsq_levels_per_channel(h, s, b, 1, -8, 7, -128, 127, 0)
        return q_pt.to(torch.int16) + q_pc.to(torch.int16)

    ref = f(x, s, b)
    out = torch.compile(f, backend="inductor", fullgraph=True)(x, s, b)
    assert out.dtype == torch.int16 and torch.equal(out, ref)
