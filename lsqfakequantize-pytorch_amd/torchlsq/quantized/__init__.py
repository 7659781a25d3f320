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



def enable_rank_sync(model, process_group=None, grads="mean"):
    """Data-parallel QAT: make every activation quantizer of `model` ONE quantizer over the batch that is sharded across the
    ranks of `process_group` (`LSQFakeQuantizer.enable_rank_sync`: observer statistics all-reduced during the init batches,
    one all-reduce of the scale / shift gradient sums per backward, replicas bit-identical).  Weight quantizers see replicated
    tensors and are left alone.  Returns the quantizers that were switched."""
    switched = []
    for mod in model.modules():
        if isinstance(mod, LSQFakeQuantizer) and mod.dtype == torch.quint8 and not mod.debug_mode:
            mod.enable_rank_sync(process_group, grads=grads)
            switched.append(mod)
    return switched


def prepare_ddp(model, process_group=None, grads="mean"):
    """`enable_rank_sync(model)` + tell DistributedDataParallel what it need not touch.  Call it on the prepared QAT model
    BEFORE wrapping it: `model = DDP(prepare_ddp(model), ...)`.

    DDP broadcasts every buffer from rank 0 at each forward (broadcast_buffers=True); for this module that is four state
    flags per quantizer, written in place -- which the module has to treat as an out-of-band write and re-read from the
    device (a host synchronisation per quantizer and step).  The flags evolve identically on every rank by construction, the
    synchronised observers' min / max too, and the synchronised quantizers' scale.grad / shift.grad arrive already reduced,
    so all of them go on the model's `_ddp_params_and_buffers_to_ignore` list.  (Parameters the quantizers create at their
    first call are only known to a DDP built AFTER that call -- as with the reference module, run one batch first.)
    `grads='ddp'`: the quantizers make no collective of their own in the LSQ steps and DDP averages scale.grad / shift.grad in
    its buckets like any other gradient (equal shards per rank; see `LSQFakeQuantizer.enable_rank_sync`) -- the parameters
    then stay OFF the ignore list."""
    synced = set(id(m) for m in enable_rank_sync(model, process_group, grads))
    ignore = list(getattr(model, "_ddp_params_and_buffers_to_ignore", []))
    for name, mod in model.named_modules():
        if not isinstance(mod, LSQFakeQuantizer):
            continue
        prefix = name + "." if name else ""
        ignore += [prefix + b for b in ("fake_quant_enabled", "observer_enabled", "learning_enabled", "current_batch")]
        if id(mod) in synced:
            if grads != "ddp":           # ('ddp': the module leaves the reduction of its gradients to DDP)
                ignore += [prefix + "scale", prefix + "shift"]
            elif getattr(mod, "scale", None) is None or getattr(mod, "shift", None) is None:
                # the quantizer creates its parameters at its FIRST call: a DDP built before that never learns of them, nobody
                # reduces their gradients in this mode, and the replicas drift apart without a sound
                import warnings
                warnings.warn("prepare_ddp(grads='ddp'): quantizer %r has not created scale / shift yet -- run one batch through the "
                              "model BEFORE wrapping it in DistributedDataParallel, or DDP will not reduce their gradients" % (name or "<root>"))
            if mod.activation_post_process is not None:
                ignore += [prefix + "activation_post_process." + b for b, _ in mod.activation_post_process.named_buffers()]
    model._ddp_params_and_buffers_to_ignore = sorted(set(ignore))
    # DDP broadcasts rank 0's parameters and buffers when it is constructed -- except what is on the list: do that once here, so
    # that replicas that did not start identical (a checkpoint restored on rank 0 only) are identical before they are left alone
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        named = dict(model.named_parameters())
        named.update(dict(model.named_buffers()))
        src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
        with torch.no_grad():
            for name in model._ddp_params_and_buffers_to_ignore:
                t = named.get(name)
                if t is not None and t.numel() > 0:
                    dist.broadcast(t.data if isinstance(t, torch.nn.Parameter) else t, src=src, group=process_group)
        for mod in model.modules():
            if isinstance(mod, LSQFakeQuantizer):
                mod._refresh_host_state()       # the flags may just have been overwritten
    return model


__all__ = ["LSQFakeQuantizer", "LSQWeightGroup", "enable_rank_sync", "prepare_ddp"] + [row[0] for row in _TABLE]
