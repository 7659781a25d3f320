"""GPU: the drop-in surface end to end -- QConfig + prepare_qat + a few optimizer steps on a small CNN."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    return torch.nn.Sequential(
        torch.nn.Conv2d(3, 16, 3, padding=1), torch.nn.ReLU(),
        torch.nn.Conv2d(16, 32, 3, padding=1, stride=2), torch.nn.ReLU(),
        torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(32, 10))


def test_qat_training_loop_with_lsq_qconfig():
    import torchlsq  # noqa: F401
    from torch.ao.quantization import QConfig, prepare_qat
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torchlsq.quantized import LSQFakeQuantizer, disable_observer, enable_fake_quant
    from torchlsq.quantized.modules.hip_observers import HipMovingAverageMinMaxObserver

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    qconfig = QConfig(
        activation=LSQFakeQuantizer.with_args(observer=MovingAverageMinMaxObserver, otype="activation", init_batches=2),
        weight=LSQFakeQuantizer.with_args(observer=MovingAveragePerChannelMinMaxObserver, otype="weight",
                                          dtype=torch.qint8, qscheme=torch.per_channel_symmetric))
    model = _model().to(dev).train()
    model.qconfig = qconfig
    prepare_qat(model, inplace=True)
    quantizers = [m for m in model.modules() if isinstance(m, LSQFakeQuantizer)]
    assert len(quantizers) >= 6                      # weight + activation quantizers of 2 convs and 1 linear
    assert any(isinstance(q.activation_post_process, HipMovingAverageMinMaxObserver) for q in quantizers)

    x = torch.randn(16, 3, 16, 16, device=dev)
    target = torch.randint(0, 10, (16,), device=dev)
    model(x)                                          # first call creates scale/shift (input passes through)
    assert all(q.scale is not None and q.scale.is_cuda for q in quantizers)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    losses = []
    for step in range(12):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(x), target)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0]
    acts = [q for q in quantizers if q.otype == 1]
    wts = [q for q in quantizers if q.otype == 0]
    assert all(int(q.current_batch[0]) == 3 and int(q.observer_enabled[0]) == 0 for q in acts)   # init phase over
    assert all(q.scale.grad is not None for q in acts + wts)
    assert all(q.shift.grad is None for q in wts) and all(q.shift.grad is not None for q in acts)
    # the model.apply helpers and the conversion-time qparams
    model.apply(disable_observer)
    model.apply(enable_fake_quant)
    for q in quantizers:
        scale, zp = q.calculate_qparams(verbose=False)
        assert (scale > 0).all() and zp.dtype == torch.int64
        assert zp.min() >= (-128 if q.otype == 0 else 0) and zp.max() <= (127 if q.otype == 0 else 255)
    # checkpoint round trip into a freshly prepared model
    sd = model.state_dict()
    model2 = _model().to(dev).train()
    model2.qconfig = qconfig
    prepare_qat(model2, inplace=True)
    model2(x)
    model2.load_state_dict(sd)
    model.eval(); model2.eval()
    assert torch.equal(model(x), model2(x))


def test_quantized_block_runs_as_a_hip_graph():
    """Steady-state QAT block (activation quantizer x per-channel weight quantizer) captured with
    torch.cuda.make_graphed_callables: forward AND backward replay as HIP graphs (possible because no op or module
    decision synchronises with the device) and equal the eager results bit for bit.  (The consumer of the two
    quantizers is a deterministic elementwise/sum expression on purpose: a convolution's weight-gradient kernel
    picks its own summation order and differs between eager and captured runs by itself.)"""
    import copy
    import torch
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torchlsq.quantized import LSQFakeQuantizer
    dev = torch.device("cuda:0")

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.aq = LSQFakeQuantizer(MovingAverageMinMaxObserver, "activation", init_batches=1)
            self.wq = LSQFakeQuantizer(MovingAveragePerChannelMinMaxObserver, "weight", dtype=torch.qint8,
                                       qscheme=torch.per_channel_symmetric)
            self.weight = torch.nn.Parameter(torch.randn(16, 16, 3, 3) * 0.05)

        def forward(self, x):
            per_channel_gain = self.wq(self.weight).sum(dim=(1, 2, 3)).view(1, -1, 1, 1)
            return self.aq(x) * per_channel_gain

    torch.manual_seed(0)
    eager = Block().to(dev).train()
    x = torch.rand(8, 16, 12, 12, device=dev)
    for _ in range(4):                                   # parameter creation + init batches (not capturable: they
        eager(x).sum().backward()                        # create parameters and flip host-side state)
    eager.zero_grad(set_to_none=True)
    assert eager.aq._h["learning"] == 1 and eager.aq._h["observer"] == 0
    graphed_src = copy.deepcopy(eager)
    sample = torch.rand(8, 16, 12, 12, device=dev, requires_grad=True)
    graphed = torch.cuda.make_graphed_callables(graphed_src, (sample,))

    for step in range(3):
        xi = torch.rand(8, 16, 12, 12, device=dev)
        gi = torch.randn(8, 16, 12, 12, device=dev)
        xa = xi.clone().requires_grad_(True)
        xb = xi.clone().requires_grad_(True)
        ya = eager(xa)
        ya.backward(gi)
        yb = graphed(xb)
        yb.backward(gi)
        assert torch.equal(ya, yb), step
        assert torch.equal(xa.grad, xb.grad), step
        for (na, pa), (nb, pb) in zip(eager.named_parameters(), graphed_src.named_parameters()):
            assert na == nb
            if na == "wq.shift":                         # symmetric: no gradient for the shift on either route
                assert pa.grad is None and pb.grad is None
            else:
                assert pa.grad is not None and torch.equal(pa.grad, pb.grad), (step, na)
        eager.zero_grad(set_to_none=True)        # (graphed callables hand out their static gradient buffers: an in-place
        graphed_src.zero_grad(set_to_none=True)  # zero would alias them -- the usual make_graphed_callables rule)


@pytest.mark.parametrize("binding", ["native", "ctypes"])
def test_eval_backward_while_the_observer_rewrites_the_parameters(binding):
    """Two calls of one quantizer before the first call's backward, during the observer-driven phase.  The module asks for
    the reference's eval backward there (x saved, mask recomputed from the parameters as they are at backward time,
    lsq_autograd.cpp:46-73): same input gradient as the reference module's own trace.  The functional's default, the saved
    one-byte mask, differentiates with the parameters the output was computed with -- both behaviours shown."""
    import importlib.util
    import json
    import os
    from torchlsq import extension as E
    from torchlsq.functional import lsq
    from torchlsq.quantized import LSQFakeQuantizer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_trace_driver", os.path.join(root, "tests", "golden", "make_module_traces.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    want = json.load(open(os.path.join(root, "tests", "golden", "module_traces.json")))["extras"]["two_calls_one_backward"]
    if binding == "native" and E.native_lsq() is None:
        pytest.skip("C++ binding not built")
    saved = E.host_binding()
    E.set_host_binding(binding)
    try:
        got = drv.two_calls_one_backward(LSQFakeQuantizer, device="cuda:0")
        for k in ("dx1_sha", "dx1_nonzero", "y1_sha", "y2_sha", "scale_after_call1", "scale_after_call2", "shift_after_call2"):
            assert got[k] == want[k], (binding, k, got[k], want[k])
        # the functional op on the same data: in-place parameter overwrite between forward and backward
        dev = torch.device("cuda:0")
        n = 4 * 8 * 6 * 6
        x = drv.S.normal_like(n, 301, 0.5, 0.2).view(4, 8, 6, 6).to(dev)
        w = drv.S.normal_like(n, 303, 0.0, 1.0).view(4, 8, 6, 6).to(dev)
        grads = {}
        for mask_backward in (True, False):
            scale = torch.tensor(want["scale_after_call1"], device=dev, requires_grad=True)
            shift = torch.tensor(want["shift_after_call1"], device=dev, requires_grad=True)
            xi = x.clone().requires_grad_(True)
            y = lsq(xi, scale, shift, 0, 127, 0, 255, eval_mode=True, mask_backward=mask_backward)
            scale.data.copy_(torch.tensor(want["scale_after_call2"], device=dev))
            shift.data.copy_(torch.tensor(want["shift_after_call2"], device=dev))
            (y * w).sum().backward()
            grads[mask_backward] = xi.grad.clone()
        assert drv.sha(grads[False].cpu()) == want["dx1_sha"]                 # the reference's answer
        assert int((grads[True] != 0).sum()) < want["dx1_nonzero"]           # forward-time mask: the extreme elements sat on the borders
        assert not torch.equal(grads[True], grads[False])
    finally:
        E.set_host_binding(saved)


def test_train_on_the_gpu_then_convert_on_the_cpu():
    """QAT on the MI355X, then the user's last step: `model.cpu()` and torch's `convert` to int8 kernels.  The quantizers'
    parameters and state follow the device move (the module re-reads its moved buffers), and the int8 model reproduces what
    the fake-quantized model computed on the GPU to within one output level."""
    import warnings
    import torchlsq  # noqa: F401
    from torch.ao.quantization import DeQuantStub, QConfig, QuantStub, convert, prepare_qat
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torchlsq.quantized import LSQFakeQuantizer

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.dq = QuantStub(), DeQuantStub()
            self.c1 = torch.nn.Conv2d(3, 16, 3, padding=1)
            self.r = torch.nn.ReLU()
            self.c2 = torch.nn.Conv2d(16, 8, 3, padding=1)

        def forward(self, x):
            return self.dq(self.c2(self.r(self.c1(self.q(x)))))

    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    m = Net().to(dev)
    m.qconfig = QConfig(activation=LSQFakeQuantizer.with_args(observer=MovingAverageMinMaxObserver, otype="activation", init_batches=3),
                        weight=LSQFakeQuantizer.with_args(observer=MovingAveragePerChannelMinMaxObserver, otype="weight", dtype=torch.qint8,
                                                          qscheme=torch.per_channel_symmetric))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        prepare_qat(m.train(), inplace=True)
        x = torch.randn(8, 3, 16, 16, device=dev)
        opt = None
        for i in range(8):
            y = m(x)
            if i == 0:
                opt = torch.optim.SGD(m.parameters(), lr=1e-3)
            opt.zero_grad()
            (y ** 2).mean().backward()
            opt.step()
        m.eval()
        y_gpu = m(x).cpu()
        m.cpu()
        y_cpu = m(x.cpu())                           # the same fake-quantized model on the CPU kernels (liblsq_cpu.so)
        mq = convert(m, inplace=False)
        y_int8 = mq(x.cpu())
    out_scale = float(mq.c2.scale)
    assert float((y_cpu - y_gpu).abs().max()) <= out_scale * 1.001          # conv arithmetic differs between the devices, the quantizers do not
    assert float((y_cpu - y_int8).abs().max()) <= out_scale * 1.001
    assert torch.equal(mq.c1.weight().int_repr(), m.c1.weight_fake_quant.quantize(m.c1.weight).int_repr())
