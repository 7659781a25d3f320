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


class HipMinMaxObserver(MinMaxObserver):
    def forward(self, x_orig):
        if not _fast(x_orig, self.min_val):
            return super().forward(x_orig)
        min_val_cur, max_val_cur = torch.ops.torchlsq.lsq_minmax_per_tensor(x_orig.detach())
        self.min_val.copy_(torch.min(min_val_cur, self.min_val))
        self.max_val.copy_(torch.max(max_val_cur, self.max_val))
        return x_orig


class HipMovingAverageMinMaxObserver(MovingAverageMinMaxObserver):
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
