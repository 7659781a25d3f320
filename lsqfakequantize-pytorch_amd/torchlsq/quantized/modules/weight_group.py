"""All weight quantizers of a model in one launch each way.

The reference runs one `LSQFakeQuantizer.forward` -> `lsq` -> per-channel kernel per layer and step
(quantized/modules/observers.py:458-461); on MI355X each of those calls is launch-latency-bound (a few MB per tensor).
`LSQWeightGroup` quantizes the weights of every `torch.ao.nn.qat` conv / linear layer of a model in ONE horizontally fused
call (`torchlsq.functional.lsq_foreach`: one launch per 32 tensors) at the start of the model's forward and hands each layer
its result when the layer asks its `weight_fake_quant` for it -- same values, same gradients (bit-identical), one autograd
node instead of dozens.

    model = prepare_qat(model)          # with LSQFakeQuantizer weight quantizers in the QConfig
    group = LSQWeightGroup(model)       # registers a forward pre-hook on `model`
    ...
    loss = model(x); loss.backward()    # weights are quantized together; the layers' own calls return the shared results

Only quantizers in their steady state take part (created, learning or plain fake-quant without observer, per-channel or
per-tensor symmetric/affine as configured, on the GPU); everything else -- the creating call, observer-driven calls,
disabled fake-quant, debug mode, CPU -- runs through the quantizer's own forward as before.
"""
import torch

from torchlsq.functional import lsq_foreach
from .observers import LSQFakeQuantizer, TYPES_RANGE_MAPPING


def _qat_weight_layers(model):
    """(layer, quantizer) pairs whose forward calls `layer.weight_fake_quant(layer.weight)` with the plain weight"""
    import torch.ao.nn.qat as nnqat
    plain = tuple(getattr(nnqat, n) for n in ("Conv1d", "Conv2d", "Conv3d", "Linear", "Embedding", "EmbeddingBag") if hasattr(nnqat, n))
    out = []
    for m in model.modules():
        if type(m) in plain and isinstance(getattr(m, "weight_fake_quant", None), LSQFakeQuantizer):
            out.append((m, m.weight_fake_quant))
    return out


class LSQWeightGroup:
    def __init__(self, model, register_hook=True):
        self.pairs = _qat_weight_layers(model)
        self.handle = model.register_forward_pre_hook(self._pre_hook) if register_hook else None
        # results a layer did not pick up (a branch that did not run, an exception mid-forward) are dropped when the model's
        # forward ends, whatever way it ends: a stashed non-leaf tensor would otherwise keep the whole fused graph alive, make
        # copy.deepcopy(model) raise and travel with torch.save(model)
        self.post_handle = model.register_forward_hook(self._post_hook, always_call=True) if register_hook else None
        self.last_fused = 0          # tensors that went through the fused call at the last prequantize()

    def clear(self):
        for _, q in self.pairs:
            q._prefetched = None

    def remove(self):
        self.clear()
        for name in ("handle", "post_handle"):
            h = getattr(self, name)
            if h is not None:
                h.remove()
                setattr(self, name, None)

    def _pre_hook(self, module, args):
        self.prequantize()

    def _post_hook(self, module, args, output):
        self.clear()

    @staticmethod
    def _steady(q, w):
        """the call `q(w)` would be a plain per-channel `lsq` call with fixed parameters: no creation, no observer, no init phase"""
        if q.debug_mode or not q._initialized or not q.is_perchannel or not w.is_cuda:
            return False
        if q._stamp != q._buffer_stamp():
            q._refresh_host_state()
        h = q._h
        if h['fake_quant'] != 1 or h['observer'] == 1:
            return False
        if h['batch'] <= q.n_batches and q.training and h['learning'] == 1:
            return False        # still inside an initialisation phase: its bookkeeping lives in forward()
        return True

    def prequantize(self):
        """Quantize every steady-state weight in one fused call and stash the results for the layers' own calls."""
        by_cfg = {}
        for layer, q in self.pairs:
            q._prefetched = None
            w = layer.weight
            if not self._steady(q, w):
                continue
            full_lsq = bool(q._h['learning'])
            q.scale.requires_grad = full_lsq
            q.shift.requires_grad = full_lsq and q.is_affine
            key = (w.device, w.dtype, q.quant_min, q.quant_max, q.dtype, q.use_grad_scaling, q.grad_scaler, q.is_affine, full_lsq)
            by_cfg.setdefault(key, []).append((q, w))
        self.last_fused = 0
        for key, items in by_cfg.items():
            if len(items) < 2:
                continue
            (_, _, qmin, qmax, qdtype, use_gs, gs, affine, full_lsq) = key
            tmin, tmax = TYPES_RANGE_MAPPING[qdtype]['range']
            ys = lsq_foreach([w for _, w in items], [q.scale for q, _ in items], [q.shift for q, _ in items], quant_min=qmin,
                             quant_max=qmax, type_min=tmin, type_max=tmax, axis=[q.ch_axis for q, _ in items],
                             use_grad_scaling=use_gs, grad_scaler=gs, is_affine=affine, eval_mode=(not full_lsq), init_mode=False)
            for (q, w), y in zip(items, ys):
                # valid for exactly this weight and these parameter values: the version counters catch an in-place update
                # (optimizer.step(), w.mul_(), ...) between prequantize() and the layer's own call
                q._prefetched = (w, w._version, q.scale._version, q.shift._version, y)
            self.last_fused += len(items)
        return self.last_fused
