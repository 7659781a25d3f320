"""`model.apply(...)` switches for fake-quant / observer state (reference torchlsq/quantized/__init__.py:5-35).

Eight callables, all the same shape -- "if the module is a fake-quantizer of the right kind, call one of its
switch methods" -- so they are generated from one table.  The `*_on_act` / `*_on_weights` variants keep the
reference's selection rule including its operator precedence: a stock `torch.quantization.FakeQuantize` always
matches, an `LSQFakeQuantizer` only when its dtype is the activation (quint8) resp. weight (qint8) type.
"""
import torch
from torch.ao.quantization import FakeQuantize as _FakeQuantize

from .modules.observers import LSQFakeQuantizer
from .modules.weight_group import LSQWeightGroup  # noqa: F401


def _switch(name, method, only_dtype=None):
    """A `model.apply` visitor calling `method()` on the fake-quantizers it is meant for."""
    def visit(mod):
        stock = isinstance(mod, _FakeQuantize)
        ours = isinstance(mod, LSQFakeQuantizer) and (only_dtype is None or mod.dtype == only_dtype)
        if stock or ours:
            getattr(mod, method)()
    visit.__name__ = visit.__qualname__ = name
    visit.__doc__ = "model.apply(%s): %s() on %s." % (
        name, method, "every fake-quantizer" if only_dtype is None else
        "stock FakeQuantize modules and LSQFakeQuantizers of dtype %s" % only_dtype)
    return visit


_TABLE = (
    # exported name,               method,               LSQFakeQuantizer dtype filter
    ("disable_fake_quant",          "disable_fake_quant", None),
    ("enable_fake_quant",           "enable_fake_quant",  None),
    ("disable_observer",            "disable_observer",   None),
    ("enable_observer",             "enable_observer",    None),
    ("disable_fake_quant_on_act",   "disable_fake_quant", torch.quint8),
    ("enable_fake_quant_on_act",    "enable_fake_quant",  torch.quint8),
    ("disable_observer_on_weights", "disable_observer",   torch.qint8),
    ("enable_observer_on_weights",  "enable_observer",    torch.qint8),
)
for _name, _method, _dtype in _TABLE:
    globals()[_name] = _switch(_name, _method, _dtype)
del _name, _method, _dtype

__all__ = ["LSQFakeQuantizer", "LSQWeightGroup"] + [row[0] for row in _TABLE]
