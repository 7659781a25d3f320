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
