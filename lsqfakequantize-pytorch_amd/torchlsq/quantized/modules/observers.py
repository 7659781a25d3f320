"""LSQFakeQuantizer -- the `torch.quantization.FakeQuantize`-compatible module around the LSQ op.

Behavioural restatement of reference torchlsq/quantized/modules/observers.py (class
`LSQFakeQuantizer`, :72-483): same constructor arguments, attributes, parameter / buffer names
(so state dicts interchange) and the same init -> observe/learn state machine in `forward`
(:424-462).  The module is the caller of the hot path; the quantize/dequantize arithmetic itself
runs in the gfx950 kernels behind `torchlsq.functional.lsq`.

State flags without host<->device round trips: the reference tests its four state buffers
(`fake_quant_enabled`, `observer_enabled`, `learning_enabled`, `current_batch`) with Python `if`s on
every forward (observers.py:431-451) -- once the module lives on the GPU each test is a blocking
device synchronisation, four per call, which serialises the whole training step.  Here the buffers stay
the persisted state (same names, dtypes and state_dict keys) but every decision reads a host-side
mirror kept by the module's own methods; the mirror is re-read from the buffers after
`load_state_dict`, on every `train()` / `eval()` call, and whenever a buffer was written behind the
module's back (`m.fake_quant_enabled[0] = 0`, `.fill_()`, `.copy_()`, a broadcast into the buffers, `.to(device)`):
`forward` compares each buffer's identity and `Tensor._version` -- host-side counters, no device access -- with
what it last saw, so such writes are honoured at the next call like in the reference, at the price of one
re-read (a synchronisation) per out-of-band write, never in the steady state.

Multi-GPU (an addition of this build; the reference has no distributed code): with `sync=True` (or `process_group=`, or
`enable_rank_sync()`) a quantizer whose input is the rank's shard of the batch -- every activation quantizer of a
data-parallel model -- behaves as ONE quantizer over the whole batch: see `enable_rank_sync`.

Two deliberate differences from the reference, both turning a crash into the documented behaviour:
  * `LSQFakeQuantizer.with_args(...)` works (the reference calls `partial` without importing it,
    observers.py:64);
  * the inner observer receives the *resolved* channel axis (0 for weights / 1 for activations when
    `ch_axis` is None) instead of the raw `None`.
"""
import inspect
from functools import partial
from math import ceil, copysign, log
from typing import Tuple

import warnings

import torch
from torch.ao.quantization.observer import ObserverBase as _TorchObserverBase

from torchlsq.functional import lsq
from .hip_observers import accelerated as _accelerated

Tensor = torch.Tensor

# ---- tables ------------------------------------------------------------------------------------
OTYPES = {'weight': 0, 'activation': 1}
TYPES_RANGE_MAPPING = {
    torch.qint8: {'range': (-128, 127), 'bitness': 8, 'unsigned': False},
    torch.quint8: {'range': (0, 255), 'bitness': 8, 'unsigned': True},
}
QSCHEMES = (torch.per_tensor_affine, torch.per_tensor_symmetric,
            torch.per_channel_affine, torch.per_channel_symmetric)
_PER_CHANNEL = (torch.per_channel_affine, torch.per_channel_symmetric)
_AFFINE = (torch.per_tensor_affine, torch.per_channel_affine)


def _known(qscheme):
    assert qscheme in QSCHEMES, f"Only following schemes supported {QSCHEMES} but recieved {qscheme}"


def IS_QSCHEME_PER_CHANNEL(qscheme):
    _known(qscheme)
    return qscheme in _PER_CHANNEL


def IS_QSCHEME_AFFINE(qscheme):
    _known(qscheme)
    return qscheme in _AFFINE


def IS_QSCHEME_PER_TENSOR(qscheme):
    return not IS_QSCHEME_PER_CHANNEL(qscheme)


def IS_QSCHEME_SYMMETRIC(qscheme):
    return not IS_QSCHEME_AFFINE(qscheme)


# ---- picklable `with_args` factories (reference :38-70) -------------------------------------------
class _PartialWrapper(object):
    def __init__(self, p):
        self.p = p

    def __call__(self, *args, **keywords):
        return self.p(*args, **keywords)

    def __repr__(self):
        return self.p.__repr__()


def _with_args(cls_or_self, **kwargs):
    """Class factory: `Foo.with_args(a=1).with_args(b=2)()` builds `Foo(a=1, b=2)`; every call of
    the factory makes a new instance (what QConfig expects of its activation/weight entries)."""
    r = _PartialWrapper(partial(cls_or_self, **kwargs))
    r.with_args = partial(_with_args, r)
    return r


class ObserverBase(_TorchObserverBase):
    with_args = classmethod(_with_args)


def _flag(value):
    return torch.tensor([int(value)], dtype=torch.uint8)


class _GroupRef(object):
    """Holds the process group of a rank-synchronised quantizer.  A ProcessGroup can neither be deep-copied nor pickled, and
    nn.Module does both with its attributes (copy.deepcopy(model), torch.save(model)): a copy shares the group, a pickle
    comes back meaning the default group."""
    def __init__(self, group=None):
        self.group = group

    def __deepcopy__(self, memo):
        return self

    def __reduce__(self):
        return (_GroupRef, ())


def _dist_world(group):
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class LSQFakeQuantizer(ObserverBase):
    """Fake quantizer with Learned Step Size Quantization (LSQ+, arXiv:2004.09576).

    Quantize -> dequantize with learnable `scale` and `shift` (see `torchlsq.functional.lsq` for
    the arithmetic).  PyTorch's quantized kernels expect qint8 weights and quint8 activations, so
    `otype` fixes the dtype: 'weight' <-> qint8 (symmetric only), 'activation' <-> quint8.

    Parameter initialisation
      * weights: statically, at the first forward, `scale = max(|mean - 3 std|, |mean + 3 std|) / 2**bits`
        (per channel for per-channel schemes);
      * activations: during the first `init_batches` training batches, either from the wrapped
        `observer` (init_mode='observer': the module acts as a plain fake-quantizer fed by the
        observer's qparams) or by back-propagating ||x_r - x||^2 into scale/shift
        (init_mode='learnable').
    The first forward only creates the parameters and returns its input unchanged -- hand the
    parameters to the optimizer after that call.

    Default ranges are 7-bit (qint8 [-64, 63], quint8 [0, 127]) to avoid overflow in PyTorch's
    quantized kernels; pass `avoid_torch_overflow=False` for the full 8 bits.

    Args:
        observer: observer *class* used for init_mode='observer' (e.g. MovingAverageMinMaxObserver).
        otype: 'weight' or 'activation'.
        dtype: torch.quint8 (default) or torch.qint8.
        qscheme: per_tensor_affine (default), per_tensor_symmetric, per_channel_affine, per_channel_symmetric.
        quant_min, quant_max: custom quantized range (both or neither).
        init_scale, init_shift: initial parameters for activations (shift is overridden for symmetric schemes).
        ch_axis: channel axis for per-channel schemes; default 0 for weights, 1 for activations.
        learn_params: learn scale/shift with LSQ (True) or behave like FakeQuantize (False).
        init_batches: number of initialisation batches for activations.
        init_mode: 'observer' or 'learnable'.
        use_grad_scaling, grad_scaler: gradient scaling of the parameters.
        avoid_torch_overflow: use 7-bit default ranges / reduce_range in the observer.
        debug_mode: forward is the identity.
        sync, process_group, sync_grads (this build): see `enable_rank_sync`; `process_group` alone implies sync=True.
    """
    init_modes = ('learnable', 'observer')
    fuse_observer_tail = True     # observer-driven init batches on the GPU: one launch after the statistics pass
    # rank sync off unless switched on (class-level defaults: a module pickled by an earlier build has no such attributes)
    _sync = False
    _sync_grads = 'sum'
    _ddp_numel = None       # sync_grads='ddp': the shard size the equal-shards shortcut was first used with
    _ddp_warned = False
    _group_ref = _GroupRef(None)

    @staticmethod
    def sign(x):
        return copysign(1, x)

    def __init__(self, observer, otype,
                 dtype=torch.quint8,
                 qscheme=torch.per_tensor_affine,
                 quant_min=None, quant_max=None,
                 init_scale=1., init_shift=0.,
                 ch_axis=None, learn_params=True,
                 init_batches=1000, init_mode='observer',
                 use_grad_scaling=True, grad_scaler=1.,
                 avoid_torch_overflow=True, debug_mode=False, sync=False, process_group=None, sync_grads='sum',
                 **observer_kwargs):
        super().__init__(dtype)
        assert init_mode in self.init_modes, f'only following modes available: {("learnable", "observer")}'
        assert otype in OTYPES, f'otype must be on of {tuple(OTYPES.keys())}, but {otype} is given'
        assert self.dtype in TYPES_RANGE_MAPPING, \
            f"Default Observer only works for {tuple(TYPES_RANGE_MAPPING.keys())} data types"
        self.otype = OTYPES[otype]
        self.qscheme = qscheme
        # channel axis: 0 for weights, 1 for activations unless given
        self.ch_axis = int(bool(self.otype)) if ch_axis is None else ch_axis

        self.activation_post_process = None
        if init_mode == 'observer':
            assert inspect.isclass(observer), 'awaited Observer class not instance or function wrapper'
            # offer the observer every constructor argument of ours that it knows by name, plus the
            # caller's extra kwargs; reduce_range mirrors avoid_torch_overflow
            offered = dict(dtype=dtype, qscheme=qscheme, quant_min=quant_min, quant_max=quant_max,
                           init_scale=init_scale, init_shift=init_shift, ch_axis=self.ch_axis,
                           learn_params=learn_params, init_batches=init_batches, init_mode=init_mode,
                           use_grad_scaling=use_grad_scaling, grad_scaler=grad_scaler,
                           avoid_torch_overflow=avoid_torch_overflow, debug_mode=debug_mode)
            offered.update(observer_kwargs)
            offered['reduce_range'] = avoid_torch_overflow
            accepted = set(inspect.signature(observer.__init__).parameters) - {'self'}
            # stock torch MinMax observers are swapped for subclasses whose statistics pass is the gfx950
            # one-pass min/max kernel (same buffers, update rules and qparams; see hip_observers.py)
            observer = _accelerated(observer)
            self.activation_post_process = observer(**{k: v for k, v in offered.items() if k in accepted})

        self.init_mode = init_mode
        self.n_batches = init_batches
        self.use_grad_scaling = use_grad_scaling
        self.grad_scaler = grad_scaler
        self.debug_mode = debug_mode
        self.is_perchannel = IS_QSCHEME_PER_CHANNEL(self.qscheme)
        self.is_affine = IS_QSCHEME_AFFINE(self.qscheme)
        self.init_scale = init_scale
        self.init_shift = init_shift
        self.quant_min, self.quant_max = self._verify_qmin_qmax(quant_min, quant_max, lowbit=avoid_torch_overflow)
        self.reset(learn_params=learn_params)
        self._sync = False
        self._sync_grads = 'sum'
        self._group_ref = _GroupRef(None)
        if sync or process_group is not None:
            self.enable_rank_sync(process_group, grads=sync_grads)

    # ---- one quantizer over a batch that is sharded across ranks (this build) ---------------------------------------
    def enable_rank_sync(self, process_group=None, grads='sum'):
        """Treat the input as this rank's shard (dim 0) of a batch spread over the ranks of `process_group` (None = the
        default group) and behave like the reference module on the WHOLE batch (quantized/modules/observers.py:424-462 on
        the concatenation of the shards), with replicated `scale` / `shift` that stay bit-identical on every rank:
          * observer-driven init batches: the batch min / max are all-reduced in ONE packed collective ([min, -max], MIN)
            before the observer's running state, the qparams and scale / shift are updated -- without it every rank
            overwrites its replica from rank-local statistics and the replicas diverge for good (DDP never re-synchronises
            parameters, only gradients);
          * LSQ steps ('learnable' init batches included): `torchlsq.distributed.lsq_sharded` -- ONE all-reduce of the fp64
            [sum d_scale terms, sum d_shift terms, element count] per backward, the gradient scaler from the global element
            count (lsq_cpu.cpp:103,250), so shards may be uneven or empty; scale.grad / shift.grad arrive already reduced
            and identical on every rank.
        `grads`: 'sum' = the gradients of the reference on the concatenated batch for the same upstream gradients;
        'mean' = that divided by the world size, DistributedDataParallel's convention (it averages every other parameter's
        gradient; DDP may average these again -- they are equal on all ranks, so that changes nothing -- or skip them:
        `torchlsq.quantized.prepare_ddp`); 'ddp' = NO collective of this module's own in the LSQ steps: the rank-local sums,
        scaled with the global element count (local numel x world size: equal shards, which DDP's averaging assumes anyway),
        are left for DDP's bucketed gradient all-reduce to average with everything else -- the same numbers as 'mean' to an
        fp32 rounding and zero extra host time per call (c10d's all_reduce enqueue is ~60 us: profiles/r04_module_sync_cost.txt),
        but only correct under a wrapper that averages gradients, the parameters must NOT be on its ignore list, and while
        observer-driven init batches are still running (the quantizer is a plain fake-quantizer then: no parameter gradients)
        DDP needs `find_unused_parameters=True`, as it does with the reference module.
        Weight quantizers see replicated tensors and need none of this: on them the switch is accepted and does nothing.
        No-op while torch.distributed is not initialised or the group has one rank."""
        assert grads in ('sum', 'mean', 'ddp'), "grads must be 'sum', 'mean' or 'ddp'"
        if self.otype != OTYPES['weight']:
            assert not (self.is_perchannel and self.ch_axis == 0), \
                'rank sync shards dim 0 (the batch): a per-channel quantizer along dim 0 has nothing to synchronise'
            obs = self.activation_post_process
            if obs is not None:
                from torch.ao.quantization.observer import MinMaxObserver, PerChannelMinMaxObserver
                assert isinstance(obs, (MinMaxObserver, PerChannelMinMaxObserver)), \
                    'rank sync needs an observer of the MinMax family (its state is a function of the batch min / max)'
        self._sync = True
        self._sync_grads = grads
        self._group_ref = _GroupRef(process_group)

    def disable_rank_sync(self):
        self._sync = False

    def _sync_world(self):
        """ranks this call synchronises with (1 = none): only quantizers of batch-sharded inputs, only inside a job"""
        if not self._sync or self.otype == OTYPES['weight']:
            return 1
        return _dist_world(self._group_ref.group)

    def _observe_synced(self, x, obs):
        """One observer step on the batch statistics of ALL ranks (see enable_rank_sync)."""
        group = self._group_ref.group
        fused = getattr(obs, 'lsq_fused_step', None) if self.fuse_observer_tail else None
        if fused is not None and fused(x, self.scale, self.shift, group=group, synced=True):
            return
        from torchlsq.distributed import all_reduce_minmax
        from torch.ao.quantization.observer import PerChannelMinMaxObserver
        per_channel = isinstance(obs, PerChannelMinMaxObserver)
        stat_dtype = obs.min_val.dtype
        if per_channel:
            n = x.shape[obs.ch_axis]
            if x.numel() > 0:
                cur_min, cur_max = torch.ops.torchlsq.lsq_minmax_per_channel(x, obs.ch_axis)
            else:
                cur_min = torch.full((n,), float('inf'), dtype=stat_dtype, device=x.device)
                cur_max = -cur_min
        elif x.numel() > 0:
            cur_min, cur_max = (t.reshape(1) for t in torch.ops.torchlsq.lsq_minmax_per_tensor(x))
        else:
            cur_min = torch.full((1,), float('inf'), dtype=stat_dtype, device=x.device)
            cur_max = -cur_min
        gmin, gmax = all_reduce_minmax(cur_min.to(stat_dtype), cur_max.to(stat_dtype), group)
        # an EMPTY batch on every rank (this rank's is: only then is it worth a look at the result) leaves the stand-in
        # extremes (+inf, -inf), which the observer would read as min = -inf, max = +inf: nothing was observed, skip the step
        if x.numel() == 0 and bool((gmin > gmax).all()):
            return
        # the stock observer only ever looks at its input through aminmax: a two-element-per-channel stand-in with the
        # global extremes drives its own update rule (running / moving average) exactly as the whole batch would
        if per_channel:
            stand_in = torch.stack([gmin, gmax]).reshape([2] + [1] * (obs.ch_axis - 1) + [gmin.numel()])
        else:
            stand_in = torch.cat([gmin, gmax])
        obs(stand_in)
        scale, zero_point = obs.calculate_qparams()
        self._set_weights(scale=scale, zero_point=zero_point)

    # ---- range bookkeeping ---------------------------------------------------------------------
    def _verify_qmin_qmax(self, quant_min: int, quant_max: int, lowbit=True) -> Tuple[int, int]:
        """Validate / derive the quantized range (reference :213-242)."""
        if self.otype == OTYPES['weight']:
            assert not self.is_affine, 'We support only symmetric scheme for weight'
            assert self.dtype == torch.qint8, \
                'Pytorch quantized operations implementaion requires `qint8` type for weights'
        else:
            assert self.dtype == torch.quint8, \
                'Pytorch quantized operations implementaion requires `quint8` type for activation'
        info = TYPES_RANGE_MAPPING[self.dtype]
        bits = info['bitness'] - int(lowbit)
        self.has_customized_qrange = (quant_min is not None) and (quant_max is not None)
        if self.has_customized_qrange:
            assert quant_min <= 0 <= quant_max, "User-specified quantization range must include 0."
            assert quant_min < quant_max, \
                "qmin must be strictly less than qmax for user-specified quantization range."
            assert 0 < quant_max - quant_min + 1 <= int(2 ** bits), \
                f"quantization range should be positive and not exceed the maximum bit range (=2^{bits})."
        else:
            quant_min, quant_max = 0, 2 ** bits - 1
            if not info['unsigned']:
                half = 2 ** (bits - 1)
                quant_min, quant_max = quant_min - half, quant_max - half
        if not self.is_affine:
            # symmetric: the shift is fixed by the (a)symmetry of the integer range
            mid = quant_min + quant_max
            self.init_shift = -float(abs(mid) // 2) * self.sign(mid) * self.init_scale
        return quant_min, quant_max

    # ---- state -----------------------------------------------------------------------------------
    @torch.jit.export
    def reset(self, learn_params=True) -> None:
        if self.otype == OTYPES['weight']:
            self.n_batches = -1          # weights are initialised statically, no init phase
        self._initialized = False
        self.register_parameter('scale', None)
        self.register_parameter('shift', None)
        self.register_buffer('fake_quant_enabled', _flag(1))
        self.register_buffer('observer_enabled', _flag(1))
        self.register_buffer('learning_enabled', _flag(learn_params))
        self.register_buffer('current_batch', torch.tensor([0], dtype=torch.int64))
        self._h = {'fake_quant': 1, 'observer': 1, 'learning': int(learn_params), 'batch': 0}   # host mirror
        self._stamp = self._buffer_stamp()
        self.enable_observer()           # applies the "observer not needed" rules below

    # ---- host mirror of the state buffers (no device synchronisation on the hot path) --------------
    def _buffer_stamp(self):
        """identity + version counter of the four state buffers: changes iff one was replaced or written in place"""
        return tuple((id(b), b._version) for b in (self.fake_quant_enabled, self.observer_enabled,
                                                   self.learning_enabled, self.current_batch))

    def _refresh_host_state(self):
        """re-read the mirror from the buffers (after load_state_dict / train() / eval() / an out-of-band write)"""
        self._h = {'fake_quant': int(self.fake_quant_enabled[0]), 'observer': int(self.observer_enabled[0]),
                   'learning': int(self.learning_enabled[0]), 'batch': int(self.current_batch[0])}
        self._stamp = self._buffer_stamp()

    def _set_flag(self, name, buffer, value):
        buffer[0] = value                 # tiny async host-to-device write, never a sync
        self._h[name] = int(value)
        self._stamp = self._buffer_stamp()

    def train(self, mode=True):
        out = super().train(mode)
        if hasattr(self, '_h'):
            self._refresh_host_state()
        return out

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._refresh_host_state()

    def check_is_init_mode(self):
        return (bool(self._h['learning']) and self.otype != OTYPES['weight']
                and self._h['batch'] <= self.n_batches)

    @torch.jit.export
    def enable_observer(self) -> None:
        needed = True
        if self._h['learning'] == 1:
            if self.otype == OTYPES['weight']:
                needed = False           # learned weights never use the observer
            elif self.init_mode == 'learnable':
                needed = False           # parameters come from back-propagation
            elif self.init_mode == 'observer' and self._h['batch'] > self.n_batches:
                needed = False           # the observer-driven init phase is over
        self._set_flag('observer', self.observer_enabled, int(needed))

    @torch.jit.export
    def disable_observer(self) -> None:
        self._set_flag('observer', self.observer_enabled, 0)

    @torch.jit.export
    def enable_fake_quant(self) -> None:
        self._set_flag('fake_quant', self.fake_quant_enabled, 1)

    @torch.jit.export
    def disable_fake_quant(self) -> None:
        self._set_flag('fake_quant', self.fake_quant_enabled, 0)

    @torch.jit.export
    def enable_param_learning(self):
        """Learn scale/shift with LSQ; static observer estimates are switched off and the
        initialisation phase is considered done."""
        self._set_flag('learning', self.learning_enabled, 1)
        self.disable_observer()
        self.n_batches = -1

    @torch.jit.export
    def enable_static_estimate(self):
        """Stop learning; scale/shift follow the observer (FakeQuantize behaviour)."""
        self._set_flag('learning', self.learning_enabled, 0)
        self.enable_observer()

    # ---- parameters ------------------------------------------------------------------------------
    def _init_weights(self, x: Tensor, _init_device=torch.device('cpu')) -> None:
        """Create `scale` / `shift` from an example input (reference :314-342)."""
        self._initialized = True
        n = x.shape[self.ch_axis] if (self.is_perchannel and x is not None) else 1
        device = x.device if x is not None else _init_device
        scale = torch.full((n,), self.init_scale, dtype=torch.float32).to(device)
        if self.otype == OTYPES['weight'] and x is not None:
            # 3-sigma rule: s = max(|mu - 3 sigma|, |mu + 3 sigma|) / 2**bits
            w = x.detach()
            other_axes = [d for d in range(w.ndim) if d != self.ch_axis]
            bits = ceil(log(self.quant_max - self.quant_min) / log(2)) - 1
            with torch.no_grad():
                if w.is_cuda and w.numel() > 0:      # one read-only pass in the gfx950 statistics kernels
                    if n == 1:
                        mu, sigma = (t.reshape(1) for t in torch.ops.torchlsq.lsq_meanstd_per_tensor(w))
                    else:
                        mu, sigma = torch.ops.torchlsq.lsq_meanstd_per_channel(w, self.ch_axis)
                elif n == 1:
                    mu, sigma = w.mean().unsqueeze(0), w.std().unsqueeze(0)
                else:
                    mu, sigma = torch.mean(w, other_axes), torch.std(w, other_axes)
                spread = torch.max(torch.abs(mu - 3 * sigma), torch.abs(mu + 3 * sigma))
                scale = (spread.to(device) / 2 ** bits).to(torch.float32)
        shift = torch.full((n,), self.init_shift, dtype=torch.float32).to(device)
        self.scale = torch.nn.Parameter(scale)
        self.shift = torch.nn.Parameter(shift)
        learn = bool(self._h['learning'])
        self.scale.requires_grad = learn
        self.shift.requires_grad = learn and self.is_affine

    def _set_weights(self, scale=None, shift=None, zero_point=None, _init_device=torch.device('cpu')):
        """Copy new values into the parameters; `zero_point` is converted to a shift
        (shift = -zero_point * scale) (reference :346-373)."""
        if self.scale is None:
            self._init_weights(None, _init_device=_init_device)   # per-tensor technical init

        def overwrite(param, value):
            with torch.no_grad():
                value = value.to(device=param.device, dtype=param.dtype)
                value.resize_(param.shape)
            param.data.copy_(value)

        if scale is not None:
            overwrite(self.scale, scale)
        if zero_point is not None:      # shift = -zero_point * scale, with the (possibly just updated) scale
            with torch.no_grad():
                shift = -zero_point.to(self.scale.device) * self.scale.detach()
        if shift is not None:
            overwrite(self.shift, shift)

    def set_weights(self, scale, zero_point=None, _init_device=torch.device('cpu')):
        self._set_weights(scale, shift=None, zero_point=zero_point, _init_device=_init_device)

    @staticmethod
    def convert_shift_to_zp(shift, scale, dtype):
        """zero_point = clamp(round(-shift / scale), type range) as int64 (reference :378-401)."""
        lo, hi = TYPES_RANGE_MAPPING[dtype]['range']
        with torch.no_grad():
            zp = -shift / scale
            zp.round_().clamp_(min=lo, max=hi)
            return zp.to(torch.int64)

    @torch.jit.export
    def calculate_qparams(self, verbose=True, need_shift=False) -> Tuple[Tensor, Tensor]:
        if not self._initialized:
            if verbose:
                print("Scale and Zero Point are not initialized properly, because  LSQObserver was never called."
                      "                       You must at least run model on random tensor, before calling convert!"
                      "                       Returned init_scale and init_zero_point")
            zp = self.convert_shift_to_zp(torch.tensor(self.init_shift), torch.tensor(self.init_scale),
                                          self.dtype).item()
            return (self.init_scale, self.init_shift, zp) if need_shift else (self.init_scale, zp)
        eps = torch.tensor(torch.finfo(torch.float32).eps)
        scale, shift = torch.max(self.scale.detach().clone().cpu(), eps), self.shift.detach().clone().cpu()
        zero_point = self.convert_shift_to_zp(shift, scale, self.dtype)
        return (scale, shift, zero_point) if need_shift else (scale, zero_point)

    def quantize(self, x):
        """`x` as a REAL quantized tensor (torch.quint8 / torch.qint8, per tensor or per channel) with this trained
        quantizer's levels and constants: `m.quantize(x).dequantize()` is `m(x)` in its steady state, bit for bit
        (`torchlsq.functional.lsq_quantize`; one pass writing one byte per element).  Note `calculate_qparams()` keeps the
        reference's formulas (max(scale, eps), round(-shift / scale)); the kernels -- and therefore this method -- use
        max(|scale|, eps) and round(clamp(-shift * (1 / s))), which differ from them only for a negative scale or when
        -shift / s sits within an ulp of a rounding tie."""
        from torchlsq.functional import lsq_quantize
        assert self._initialized and self.scale is not None, "run the module on at least one batch before quantize()"
        tmin, tmax = TYPES_RANGE_MAPPING[self.dtype]['range']
        return lsq_quantize(x, self.scale, self.shift, quant_min=self.quant_min, quant_max=self.quant_max, type_min=tmin,
                            type_max=tmax, axis=self.ch_axis, is_perchannel=self.is_perchannel, dtype=self.dtype)

    # ---- the caller of the hot path ------------------------------------------------------------
    _prefetched = None      # (weight, its version, scale's, shift's, fake-quantized value) stashed by LSQWeightGroup.prequantize()

    def forward(self, x):
        pre = self._prefetched
        if pre is not None:         # this call's result was computed with the other weight quantizers, in one launch
            self._prefetched = None
            # ... for THIS tensor object and the values it and the parameters had then (Tensor._version: host-side counters)
            if pre[0] is x and pre[1] == x._version and pre[2] == self.scale._version and pre[3] == self.shift._version:
                return pre[4]
        if self.debug_mode:
            return x
        if not self._initialized:
            self._init_weights(x)
            return x                        # the creating call passes its input through
        if self._stamp != self._buffer_stamp():
            self._refresh_host_state()       # a state buffer was written or replaced outside the module's methods
        h = self._h                          # host mirror: no device synchronisation below
        full_lsq = bool(h['learning'])
        backprop_init = False
        if h['batch'] <= self.n_batches and self.training and h['learning'] == 1:
            last = h['batch'] == self.n_batches
            if self.init_mode == 'observer':
                # plain fake-quant driven by the observer until the last init batch
                full_lsq = last
                if last:
                    self.disable_observer()
            elif self.init_mode == 'learnable':
                self.disable_observer()
                backprop_init = not last
            self.current_batch[0] += 1       # in place on the device, asynchronous
            h['batch'] += 1
            self._stamp = self._buffer_stamp()

        sync_ws = self._sync_world()
        if h['observer'] == 1:
            obs = self.activation_post_process
            fused = getattr(obs, 'lsq_fused_step', None) if self.fuse_observer_tail else None
            # GPU fast path: statistics pass + ONE launch that updates the observer state, derives the qparams and
            # writes scale / shift, nothing read back to the host (hip_observers.py); otherwise the reference sequence
            if sync_ws > 1:
                self._observe_synced(x.detach(), obs)
            elif fused is None or not fused(x.detach(), self.scale, self.shift):
                obs(x.detach())
                scale, zero_point = obs.calculate_qparams()
                self._set_weights(scale=scale, zero_point=zero_point)

        if h['fake_quant'] == 1:
            backprop_init = bool(backprop_init and full_lsq)
            tmin, tmax = TYPES_RANGE_MAPPING[self.dtype]['range']
            self.scale.requires_grad = full_lsq
            self.shift.requires_grad = full_lsq and self.is_affine
            # While the observer is enabled it overwrites scale / shift in place on EVERY call (above), so a second call
            # before the first one's backward changes what the reference's eval backward sees (it recomputes the mask
            # from the saved x and the then-current parameters, lsq_autograd.cpp:46-73): keep that behaviour there
            # (save x); once the parameters are only changed by the optimizer, the one-byte saved mask is equivalent.
            if sync_ws > 1 and full_lsq:
                # the input is this rank's shard of the batch: one all-reduce per backward, scaler from the global count
                from torchlsq.distributed import COLLECTIVE, lsq_sharded
                ddp = self._sync_grads == 'ddp'       # the wrapper averages the gradients: no collective of our own
                if ddp:
                    # ... with the GLOBAL count taken as local numel x world size: right for equal shards only, and this rank
                    # cannot know the others' sizes without the collective this mode exists to avoid (nor may it enter one on
                    # its own: the other ranks would not).  A shard unlike the first one (the last batch of an epoch) gets a
                    # gradient scaler that is off by sqrt(its share): say so, once.
                    if self._ddp_numel is None:
                        self._ddp_numel = x.numel()
                    elif x.numel() != self._ddp_numel and not self._ddp_warned:
                        self._ddp_warned = True
                        warnings.warn("LSQFakeQuantizer(sync_grads='ddp') assumes equal shards on every rank: this call's shard has %d "
                                      "elements, the first one had %d -- the gradient scaler 1/sqrt(numel * quant_max) of this step uses "
                                      "%d x world size as the batch's element count.  Use sync_grads='mean' (prepare_ddp's default: the "
                                      "count travels in the collective) or drop the ragged last batch." % (x.numel(), self._ddp_numel, x.numel()))
                gs = self.grad_scaler / sync_ws if self._sync_grads == 'mean' else self.grad_scaler
                return lsq_sharded(x, self.scale, self.shift, quant_min=self.quant_min, quant_max=self.quant_max,
                                   type_min=tmin, type_max=tmax, axis=self.ch_axis, use_grad_scaling=self.use_grad_scaling,
                                   grad_scaler=gs, is_affine=self.is_affine, is_perchannel=self.is_perchannel,
                                   eval_mode=False, init_mode=backprop_init, group=self._group_ref.group,
                                   global_numel=None if ddp else COLLECTIVE, reduce=not ddp)
            return lsq(x, self.scale, self.shift, quant_min=self.quant_min, quant_max=self.quant_max,
                       type_min=tmin, type_max=tmax, axis=self.ch_axis, use_grad_scaling=self.use_grad_scaling,
                       grad_scaler=self.grad_scaler, is_affine=self.is_affine, is_perchannel=self.is_perchannel,
                       eval_mode=(not full_lsq), init_mode=backprop_init, mask_backward=(h['observer'] != 1))
        return x

    @torch.jit.export
    def extra_repr(self):
        if self.debug_mode:
            return 'Debug mode: ON, doing nothing.'
        scale, shift, zp = self.calculate_qparams(verbose=False, need_shift=True)
        head = '' if self._initialized else '(Uninitialized!) '
        if self.check_is_init_mode():
            head += (f'(Observer in parameter init mode: {self.init_mode}; '
                     f'{self.current_batch[0]}/{self.n_batches} batches left) ')
        target = 'weights' if self.otype == OTYPES['weight'] else 'activation'
        per_channel = f'Yes, channel axis - {self.ch_axis}' if self.is_perchannel else 'No'
        torch.set_printoptions(threshold=8)
        text = (f"{head}Observer for {target}; Learnable:{bool(self.learning_enabled[0])}; "
                f"Observer:{bool(self.observer_enabled[0])}; FakeQuant:{bool(self.fake_quant_enabled[0])}; "
                f"Qtype:{self.dtype}, Affine:{self.is_affine}, PerChannel:{per_channel}, "
                f"Qrange:[{self.quant_min},{self.quant_max}], scale={scale}, zero_point={zp} (shift={shift}).")
        if hasattr(self, 'recalibrated'):
            text += '\nModule was recalibrated!'
        torch.set_printoptions(threshold=1000)
        return text
