"""MinMax observers whose statistics pass runs on the gfx950 one-pass min/max kernel.

During its initialisation batches LSQFakeQuantizer feeds every input to a torch MinMax observer right
before the fake-quantize op (reference quantized/modules/observers.py:446-449).  The stock observers
call torch.aminmax -- and the per-channel ones first permute + flatten the input, a full copy of x.
The subclasses below keep the stock classes' state (buffers, state_dict keys, calculate_qparams) and
update rules verbatim and replace only the reduction, for GPU tensors, with
`torch.ops.torchlsq.lsq_minmax_per_tensor / _per_channel` (one read-only pass over HBM; torch.aminmax
semantics incl. NaN propagation).  CPU tensors take the stock code path.

LSQFakeQuantizer swaps a stock class for its subclass automatically (`accelerated(observer_cls)`), so
user code keeps passing `MovingAverageMinMaxObserver` etc.

`lsq_fused_step` goes one step further for the module's initialisation batches: after the statistics pass ONE more
launch (`lsq_hip_observer_update`) updates the observer's running state, derives torch's qparams from it and
writes the module's `scale` / `shift` parameters -- instead of the stock sequence "update buffers, calculate_qparams,
_set_weights", which is a dozen tiny tensor kernels and several host synchronisations (`if min_val == inf`,
`check_min_max_valid`, `float(scale)`) per call.  Same numbers, bit for bit (tests/test_parity_gpu.py).
"""
import torch
from torch.ao.quantization.observer import (MinMaxObserver, MovingAverageMinMaxObserver,
                                            MovingAveragePerChannelMinMaxObserver, PerChannelMinMaxObserver)

_FAST_DTYPES = (torch.float32, torch.float64, torch.bfloat16, torch.float16)


def _fast(x, buf):
    """the kernel applies: GPU tensor, supported storage type, and the observer's buffer has the type
    the kernel reports (fp32 for <= 32-bit inputs, fp64 for fp64) -- otherwise fall back to the stock path."""
    if not (x.is_cuda and x.dtype in _FAST_DTYPES and x.numel() > 0):
        return False
    want = torch.float64 if x.dtype == torch.float64 else torch.float32
    return buf.dtype == want


def _fused_step(obs, x, scale_param, shift_param, mode, per_channel, group=None, synced=False):
    """statistics pass + one update launch; False when the stock path must be taken instead.
    synced: the batch is sharded over the ranks of `group` -- the batch min / max of all ranks are combined in ONE packed
    collective between the two launches (torchlsq.distributed.all_reduce_minmax), so every rank writes the same scale / shift;
    an empty local shard contributes (+inf, -inf)."""
    from torchlsq import extension as E
    empty = synced and x.numel() == 0 and x.is_cuda and x.dtype in _FAST_DTYPES
    if not ((empty or _fast(x, obs.min_val)) and obs.min_val.dtype == torch.float32 and scale_param.dtype == torch.float32
            and obs.qscheme in (torch.per_tensor_affine, torch.per_tensor_symmetric, torch.per_channel_affine,
                                torch.per_channel_symmetric)):
        return False
    n = x.shape[obs.ch_axis] if per_channel else 1
    if scale_param.numel() != n or shift_param.numel() != n or not (scale_param.is_cuda and shift_param.is_cuda):
        return False
    if empty:
        cur_min = torch.full((n,), float("inf"), dtype=torch.float32, device=x.device)
        cur_max = -cur_min
    if per_channel:
        if not empty:
            cur_min, cur_max = torch.ops.torchlsq.lsq_minmax_per_channel(x.detach(), obs.ch_axis)
        if synced:
            from torchlsq.distributed import all_reduce_minmax
            cur_min, cur_max = all_reduce_minmax(cur_min, cur_max, group)
        first = 0
        if obs.min_val.numel() == 0 or obs.max_val.numel() == 0:      # host-side fact: the buffers are still empty
            obs.min_val.resize_(cur_min.shape)
            obs.max_val.resize_(cur_max.shape)
            first = 1
    else:
        if not empty:
            cur_min, cur_max = torch.ops.torchlsq.lsq_minmax_per_tensor(x.detach())
            cur_min, cur_max = cur_min.reshape(1), cur_max.reshape(1)
        if synced:
            from torchlsq.distributed import all_reduce_minmax
            cur_min, cur_max = all_reduce_minmax(cur_min, cur_max, group)
        first = -1                                                     # +inf / -inf state: decided on the device
    eps = getattr(obs, "_lsq_eps", None)
    if eps is None:                                                    # read the observer's eps buffer once
        eps = obs._lsq_eps = float(obs.eps)
    symmetric = obs.qscheme in (torch.per_tensor_symmetric, torch.per_channel_symmetric)
    zp_sym = 0
    if obs.dtype in (torch.quint8, torch.uint8):
        zp_sym = (obs.quant_min + obs.quant_max) // 2 if obs.has_customized_qrange else 128
    min_state = obs.min_val if obs.min_val.dim() else obs.min_val.view(1)
    max_state = obs.max_val if obs.max_val.dim() else obs.max_val.view(1)
    E.hip_observer_update(cur_min, cur_max, min_state, max_state, scale_param.data, shift_param.data, mode, first,
                          getattr(obs, "averaging_constant", 0.0), obs.quant_min, obs.quant_max, symmetric, zp_sym, eps)
    return True


class HipMinMaxObserver(MinMaxObserver):
    def lsq_fused_step(self, x, scale_param, shift_param, group=None, synced=False):
        return _fused_step(self, x, scale_param, shift_param, 1, False, group, synced)

    def forward(self, x_orig):
        if not _fast(x_orig, self.min_val):
            return super().forward(x_orig)
        min_val_cur, max_val_cur = torch.ops.torchlsq.lsq_minmax_per_tensor(x_orig.detach())
        self.min_val.copy_(torch.min(min_val_cur, self.min_val))
        self.max_val.copy_(torch.max(max_val_cur, self.max_val))
        return x_orig


class HipMovingAverageMinMaxObserver(MovingAverageMinMaxObserver):
    def lsq_fused_step(self, x, scale_param, shift_param, group=None, synced=False):
        return _fused_step(self, x, scale_param, shift_param, 2, False, group, synced)

    def forward(self, x_orig):
        if not _fast(x_orig, self.min_val):
            return super().forward(x_orig)
        min_val_cur, max_val_cur = torch.ops.torchlsq.lsq_minmax_per_tensor(x_orig.detach())
        min_val, max_val = self.min_val, self.max_val
        if min_val == float("inf") and max_val == float("-inf"):
            min_val, max_val = min_val_cur, max_val_cur
        else:
            min_val = min_val + self.averaging_constant * (min_val_cur - min_val)
            max_val = max_val + self.averaging_constant * (max_val_cur - max_val)
        self.min_val.copy_(min_val)
        self.max_val.copy_(max_val)
        return x_orig


class HipPerChannelMinMaxObserver(PerChannelMinMaxObserver):
    def lsq_fused_step(self, x, scale_param, shift_param, group=None, synced=False):
        return _fused_step(self, x, scale_param, shift_param, 1, True, group, synced)

    def _forward(self, x_orig):
        if not _fast(x_orig, self.min_val):
            return super()._forward(x_orig)
        min_val_cur, max_val_cur = torch.ops.torchlsq.lsq_minmax_per_channel(x_orig.detach(), self.ch_axis)
        min_val, max_val = self.min_val, self.max_val
        if min_val.numel() == 0 or max_val.numel() == 0:
            min_val, max_val = min_val_cur, max_val_cur
        else:
            min_val = torch.min(min_val_cur, min_val)
            max_val = torch.max(max_val_cur, max_val)
        self.min_val.resize_(min_val.shape)
        self.max_val.resize_(max_val.shape)
        self.min_val.copy_(min_val)
        self.max_val.copy_(max_val)
        return x_orig


class HipMovingAveragePerChannelMinMaxObserver(MovingAveragePerChannelMinMaxObserver):
    def lsq_fused_step(self, x, scale_param, shift_param, group=None, synced=False):
        return _fused_step(self, x, scale_param, shift_param, 2, True, group, synced)

    def forward(self, x_orig):
        if not _fast(x_orig, self.min_val):
            return super().forward(x_orig)
        min_val_cur, max_val_cur = torch.ops.torchlsq.lsq_minmax_per_channel(x_orig.detach(), self.ch_axis)
        min_val, max_val = self.min_val, self.max_val
        if min_val.numel() == 0 or max_val.numel() == 0:
            min_val, max_val = min_val_cur, max_val_cur
        else:
            min_val = min_val + self.averaging_constant * (min_val_cur - min_val)
            max_val = max_val + self.averaging_constant * (max_val_cur - max_val)
        self.min_val.resize_(min_val.shape)
        self.max_val.resize_(max_val.shape)
        self.min_val.copy_(min_val)
        self.max_val.copy_(max_val)
        return x_orig


_ACCELERATED = {
    MinMaxObserver: HipMinMaxObserver,
    MovingAverageMinMaxObserver: HipMovingAverageMinMaxObserver,
    PerChannelMinMaxObserver: HipPerChannelMinMaxObserver,
    MovingAveragePerChannelMinMaxObserver: HipMovingAveragePerChannelMinMaxObserver,
}


def accelerated(observer_cls):
    """the drop-in subclass for a stock MinMax observer class; anything else is returned unchanged."""
    return _ACCELERATED.get(observer_cls, observer_cls)
