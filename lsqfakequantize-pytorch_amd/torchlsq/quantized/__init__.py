"""`model.apply(...)` switches for fake-quant / observer state (reference torchlsq/quantized/__init__.py:5-35).

The `*_on_act` / `*_on_weights` helpers keep the reference's exact selection rule, including its
operator precedence: a stock `torch.quantization.FakeQuantize` always matches, an `LSQFakeQuantizer`
only when its dtype is the activation (quint8) resp. weight (qint8) type.
"""
import torch
from torch.ao.quantization import FakeQuantize as _FakeQuantize

from .modules.observers import LSQFakeQuantizer


def _is_fake_quant(mod):
    return isinstance(mod, (_FakeQuantize, LSQFakeQuantizer))


def _matches(mod, qdtype):
    return isinstance(mod, _FakeQuantize) or (isinstance(mod, LSQFakeQuantizer) and mod.dtype == qdtype)


def disable_fake_quant(mod):
    if _is_fake_quant(mod):
        mod.disable_fake_quant()


def enable_fake_quant(mod):
    if _is_fake_quant(mod):
        mod.enable_fake_quant()


def disable_observer(mod):
    if _is_fake_quant(mod):
        mod.disable_observer()


def enable_observer(mod):
    if _is_fake_quant(mod):
        mod.enable_observer()


def disable_fake_quant_on_act(mod):
    if _matches(mod, torch.quint8):
        mod.disable_fake_quant()


def enable_fake_quant_on_act(mod):
    if _matches(mod, torch.quint8):
        mod.enable_fake_quant()


def disable_observer_on_weights(mod):
    if _matches(mod, torch.qint8):
        mod.disable_observer()


def enable_observer_on_weights(mod):
    if _matches(mod, torch.qint8):
        mod.enable_observer()
