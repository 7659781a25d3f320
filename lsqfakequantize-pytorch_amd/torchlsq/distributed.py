"""Batch-sharded LSQ across the GPUs of one node: one process per GPU, ONE collective per backward.

The reference has no distributed code.  The hot path is elementwise
plus a reduction, so it shards by splitting dim 0 (the batch) across ranks:
  * forward and dx need no communication;
  * d_scale / d_shift are sums over all elements, so each rank reduces its shard to the *un-rounded*
    fp64 pair [sum ds terms, sum db terms] (per-tensor: 2 doubles; per-channel: 2*C doubles), and a
    single RCCL all-reduce(SUM) over xGMI combines them -- 16 bytes for the per-tensor case, i.e. a
    latency-bound message: one packed call, launched on the same stream right behind the backward
    kernel, no host synchronisation;
  * the gradient scaler 1/sqrt(numel*quant_max) uses the GLOBAL element count, so the result equals
    the reference's on the concatenated (unsharded) tensor (lsq_cpu.cpp:103 uses x.numel()).
`backend="nccl"` is RCCL on ROCm builds of PyTorch; the CPU tests use gloo.
"""
import os
import time

import torch
import torch.distributed as dist

from . import _abi
from . import extension as _E
from .extension import _assert_has_ops, _param_dtype


_ASSUME_PEERS = [False]     # measurements / tests on ONE GPU: communicate as if the process group had more than one rank


def _world(group):
    ws = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    return max(ws, 2) if _ASSUME_PEERS[0] else ws


def assume_peers(on=True):
    """A world of one that communicates as if it had peers: every all-reduce is then an identity, so the sharded op must
    equal the plain op while the whole call path -- extra launches, the collective's enqueue -- is the N > 1 one.  For the
    one-GPU measurements and tests (bench.py `cfg4_shard_collective`, tests/test_rccl_world1_gpu.py)."""
    _ASSUME_PEERS[0] = bool(on)


COLLECTIVE = "collective"      # global_numel=COLLECTIVE: the element count travels in the all-reduce (uneven shards)


# ---- the collective itself: the library's own RCCL communicator when there is one, torch.distributed otherwise ----------
# torch.distributed's all_reduce costs ~60 us of HOST time per call (profiles/r04_module_sync_cost.txt), as much as a rank's
# whole BASELINE-config-4 step takes on the GPU; lsq_hip_comm_all_reduce* (include/lsq_hip.h) makes the same RCCL call with a
# handful of HIP calls.  One communicator per (process group, GPU), created at the first sharded backward of GPU tensors over
# an RCCL ("nccl") process group: rank 0's 128-byte id is broadcast through that group, every rank joins, and the ranks then
# AGREE (one MIN all-reduce of a flag) that all of them succeeded -- otherwise all of them keep torch.distributed.
# TORCHLSQ_COLLECTIVE=c10d switches the native route off; gloo groups and CPU tensors never take it.
_COMMS = {}                 # (group key, device index) -> HipComm, or None: this group stays on torch.distributed
_NATIVE_COLLECTIVE = [os.environ.get("TORCHLSQ_COLLECTIVE", "native").lower() != "c10d"]


def set_native_collective(on):
    """True: GPU tensors over an RCCL group reduce through the library's own communicator (the default); False: always
    torch.distributed.all_reduce.  Communicators already created stay alive but are not used while it is off."""
    _NATIVE_COLLECTIVE[0] = bool(on)


def _group_key(group):
    return 0 if group is None else id(group)


def native_comm(group, device, create=True):
    """The HipComm of (group, device), created on first use -- a COLLECTIVE call then: every rank of the group must make its
    first sharded call at the same point, which data-parallel training does by construction.  None = torch.distributed.
    A communicator is only kept if it WORKS: right after creation the ranks add up rank + 1 through it under a host-side
    deadline (TORCHLSQ_COMM_CHECK_S, 60 s) and agree on the outcome; a rank that could not create it, a wrong sum or a
    reduction that does not finish leaves every rank on torch.distributed."""
    if not _NATIVE_COLLECTIVE[0] or not (dist.is_available() and dist.is_initialized()):
        return None
    key = (_group_key(group), device.index)
    if key in _COMMS:
        return _COMMS[key]
    if not create or dist.get_backend(group) != "nccl":
        _COMMS[key] = None
        return None
    rank, ws = dist.get_rank(group), dist.get_world_size(group)
    comm, ok = None, 1
    try:
        uid = torch.zeros(_E.LSQ_COMM_ID_BYTES, dtype=torch.uint8, device=device)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(_E.HipComm.unique_id()), dtype=torch.uint8))
    except Exception:       # no RCCL behind the library: still take part in the two collectives below
        ok = 0
        uid = torch.zeros(_E.LSQ_COMM_ID_BYTES, dtype=torch.uint8, device=device)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(uid, src=src, group=group)
    hung = 0
    if ok:
        try:
            comm = _E.HipComm(bytes(uid.cpu().numpy().tobytes()), rank, ws, device)
            # ... and it has to WORK before anything relies on it: the ranks add up rank + 1 through it, on a stream of its
            # own (a reduction that never ends must not sit in the caller's stream) and under a host-side deadline
            pre = torch.cuda.Stream(device=device)
            with torch.cuda.stream(pre):
                probe = torch.full((2,), float(rank + 1), dtype=torch.float64, device=device)
                comm.all_reduce(probe)
                ev = torch.cuda.Event()
                ev.record(pre)
            limit = float(os.environ.get("TORCHLSQ_COMM_CHECK_S", "60"))       # (tests: a negative limit = "it never finished")
            deadline = time.monotonic() + limit
            while not ev.query() and time.monotonic() < deadline:
                time.sleep(0.002)
            if limit < 0 or not ev.query():
                ok, hung = 0, 1
            elif float(probe[0].item()) != ws * (ws + 1) / 2.0:
                ok = 0
        except Exception:
            ok = 0
    flag = torch.tensor([ok, -hung], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag[0].item()) != 1:
        if comm is not None and int(flag[1].item()) == 0:       # (hung anywhere: left alone, a teardown would wait for it too)
            try:
                comm.destroy()
            except Exception:
                pass
        comm = None
    _COMMS[key] = comm
    return comm


def destroy_native_comms():
    """tear the library's communicators down (before dist.destroy_process_group(); collective per communicator)"""
    for key, comm in list(_COMMS.items()):
        if comm is not None:
            comm.destroy()
        del _COMMS[key]


class _NativeWork:
    """What sharded_backward(async_op=True) hands back on the native route.  The reduction runs on the communicator's own
    stream and so does its first consumer -- `rounded`: the sums cast to the parameter type there -- so the compute stream
    never waits for a single reduction: wait() makes the current stream wait (not the host) for THIS reduction and every
    earlier one of the communicator (its stream is in order), and a training step needs one such join, before the
    optimizer reads the gradients -- not one per quantizer.  (A cross-stream wait costs the GPU ~7 us each way,
    profiles/r05_comm_cost.txt: per reduction that is 14 % of a BASELINE-config-4 shard step, per step it is noise.)"""
    __slots__ = ("comm", "ticket", "rounded", "deferred")

    def __init__(self, comm, ticket, rounded=None):
        self.comm, self.ticket, self.rounded, self.deferred = comm, ticket, rounded, True

    def wait(self):
        if self.rounded is not None:
            self.comm.join()             # the reduction AND the rounding behind it (and everything earlier on that stream)
            self.rounded.record_stream(torch.cuda.current_stream(self.rounded.device))
        else:
            self.comm.end(self.ticket)
        return True


def join(group=None, device=None):
    """The once-per-step join of the native route: the current stream of `device` waits (the host does not) for every
    reduction begun on the library's communicator of (group, device) and for the rounding placed behind it -- call it before
    the optimizer reads gradients that came from `sharded_backward(..., async_op=True)`.  A no-op when the group reduces
    through torch.distributed (its Work handles are waited individually)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    comm = native_comm(group, dev, create=False)
    if comm is not None:
        comm.join()


def _all_reduce_sum(t, group, async_op=False, round_to=None):
    """in-place SUM of the fp64 buffer `t` over the ranks: the one collective of a sharded backward.
    async_op on the native route: the reduction and (round_to: a dtype) the rounding of its result run on the communicator's
    stream; see _NativeWork."""
    comm = native_comm(group, t.device) if t.is_cuda else None
    if comm is not None:
        if async_op:
            ticket = comm.begin(t)
            rounded = None
            if round_to is not None:
                side = comm.side_stream()
                with torch.cuda.stream(side):
                    rounded = t.to(round_to)
                t.record_stream(side)
            return _NativeWork(comm, ticket, rounded)
        comm.all_reduce(t)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def _finish(packed, channels, per_channel, x_dtype, qmax, use_gs, gs):
    fn = _E.hip_sharded_finish if packed.is_cuda else _E.cpu_sharded_finish
    return fn(packed, channels, per_channel, x_dtype, qmax, use_gs, gs)


def _backward_counted(grad, x, scale, shift, quant_min, quant_max, type_min, type_max, axis, use_grad_scaling, grad_scaler,
                      sym, is_perchannel, init_mode, group, ws):
    """The sharded backward when no rank knows the global element count (include/lsq_hip.h, lsq_hip_sharded_finish): the local
    kernel leaves its terms unscaled, the count rides in the last slot of the ONE all-reduced buffer, and the scaler is
    derived from the summed count on the device -- no extra collective, no host synchronisation."""
    C = scale.numel() if is_perchannel else 1
    n_local = x.numel()
    if x.is_cuda and _E._NATIVE_LSQ is not None:
        # the C++ host binding: two host calls around the collective instead of four through ctypes (~25 us less host time per
        # synchronised quantizer and backward: profiles/r05_module_sync_cost.txt)
        ops = torch.ops.torchlsq_native
        dx, packed = ops.lsq_backward_packed(grad, x, scale, shift, is_perchannel, axis, quant_min, quant_max, type_min, type_max, sym,
                                             init_mode)
        if ws > 1:
            _all_reduce_sum(packed, group)
        ds, db = ops.lsq_sharded_finish(packed, C, is_perchannel, _abi._DTYPE_CODE[x.dtype], quant_max, use_grad_scaling, grad_scaler)
        return dx, ds, db
    packed = torch.full((2 * C + 1,), float(n_local), dtype=torch.float64, device=x.device)   # slot 2C = this shard's count
    if x.is_cuda:
        if is_perchannel:
            dx, _ = _E.hip_backward_per_channel(grad, x, scale, shift, axis, quant_min, quant_max, type_min, type_max, False,
                                                1.0, sym, False, init_mode, want_wide=True, wide_out=packed)
        else:
            dx, _ = _E.hip_backward_per_tensor(grad, x, scale, shift, quant_min, quant_max, type_min, type_max, False, 1.0,
                                               sym, False, init_mode, want_wide=True, wide_out=packed)
    else:
        dx, _ = _E.cpu_backward(grad, x, scale, shift, axis, is_perchannel, quant_min, quant_max, type_min, type_max, False,
                                1.0, sym, False, init_mode, want_wide=True, wide_out=packed)
    if ws > 1:
        _all_reduce_sum(packed, group)
    ds, db = _finish(packed, C, is_perchannel, x.dtype, quant_max, use_grad_scaling, grad_scaler)
    return dx, ds, db


def sharded_backward(grad, x, scale, shift, quant_min, quant_max, type_min, type_max, axis=1,
                     use_grad_scaling=True, grad_scaler=1.0, is_affine=True, is_perchannel=False,
                     eval_mode=False, init_mode=False, group=None, global_numel=None, async_op=False, reduce=True):
    """Local fused backward + the one all-reduce.  Returns (dx, ds, db[, work]).

    reduce=False: no collective here -- the rank-local sums (already scaled with the GLOBAL element count) are returned and
    something else adds them up over the ranks: DistributedDataParallel's own bucketed gradient all-reduce, which averages
    scale.grad / shift.grad together with every other parameter's gradient.  Needs the global count up front.

    `global_numel`: None = local numel * world size (equal shards); an int = the caller knows it; COLLECTIVE = nobody
    does (uneven shards): the count is summed in the same collective and the scaler derived from it on the device.
    eval_mode: d_scale = d_shift = 0 (lsq_kernel.h:142-144), so nothing is communicated."""
    ws = _world(group)
    sym = not is_affine
    if global_numel == COLLECTIVE and not eval_mode:
        assert not async_op, "async_op is not available with global_numel=COLLECTIVE"
        assert reduce, "reduce=False needs the global element count up front (global_numel=None or an int)"
        return _backward_counted(grad, x, scale, shift, quant_min, quant_max, type_min, type_max, axis, use_grad_scaling,
                                 grad_scaler, sym, is_perchannel, init_mode, group, ws)
    n4s = x.numel() * ws if global_numel is None or global_numel == COLLECTIVE else int(global_numel)
    # GPU tensors go through the C++ host binding when it is loaded (same C ABI, same kernels; ~4x less host time per
    # call than the Python-registered op, which matters when a rank's shard is only tens of microseconds of GPU work)
    ops = torch.ops.torchlsq_native if (x.is_cuda and _E._NATIVE_LSQ is not None) else torch.ops.torchlsq
    if async_op and ws > 1 and reduce and not eval_mode and ops is not torch.ops.torchlsq and x.numel() > 0:
        # native collective + native host binding: backward, reduction on the communicator's stream and the rounding behind it
        # in ONE host call (csrc/torch_binding: lsq_backward_*_sharded)
        comm = native_comm(group, x.device)
        if comm is not None:
            if is_perchannel:
                dx, wide, rounded, ticket = ops.lsq_backward_per_channel_sharded(grad, x, scale, shift, axis, quant_min, quant_max, type_min,
                                                                                 type_max, use_grad_scaling, grad_scaler, sym, eval_mode,
                                                                                 init_mode, n4s, comm.handle)
            else:
                dx, wide, rounded, ticket = ops.lsq_backward_per_tensor_sharded(grad, x, scale, shift, quant_min, quant_max, type_min, type_max,
                                                                                use_grad_scaling, grad_scaler, sym, eval_mode, init_mode, n4s,
                                                                                comm.handle)
            return dx, wide, _NativeWork(comm, ticket, rounded)
    if is_perchannel:
        dx, wide = ops.lsq_backward_per_channel_wide(grad, x, scale, shift, axis, quant_min, quant_max, type_min,
                                                     type_max, use_grad_scaling, grad_scaler, sym, eval_mode, init_mode,
                                                     n4s)
    else:
        dx, wide = ops.lsq_backward_per_tensor_wide(grad, x, scale, shift, quant_min, quant_max, type_min, type_max,
                                                    use_grad_scaling, grad_scaler, sym, eval_mode, init_mode, n4s)
    work = None
    if ws > 1 and not eval_mode and reduce:
        work = _all_reduce_sum(wide, group, async_op=async_op, round_to=_param_dtype(x) if async_op else None)
    pd = _param_dtype(x)
    if async_op and work is not None:
        # caller waits (work.wait(): a stream-level wait), then rounds: wide[0].to(pd), wide[1].to(pd) -- or, native route
        # (getattr(work, "deferred", False)), takes work.rounded and joins once per step instead of once per reduction
        return dx, wide, work
    ds = wide[0].to(pd).reshape(-1)
    db = wide[1].to(pd).reshape(-1)
    return dx, ds, db


class _ShardedLSQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift, cfg):
        (qmin, qmax, tmin, tmax, axis, use_gs, gs, is_affine, is_pc, eval_mode, init_mode, group, gnumel, reduce) = cfg
        ops = torch.ops.torchlsq_native if (x.is_cuda and _E._NATIVE_LSQ is not None) else torch.ops.torchlsq
        sym = not is_affine
        if is_pc:
            y = ops.lsq_forward_per_channel(x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode,
                                            init_mode)
        else:
            y = ops.lsq_forward_per_tensor(x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode)
        ctx.save_for_backward(x, scale, shift)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, grad_out):
        x, scale, shift = ctx.saved_tensors
        (qmin, qmax, tmin, tmax, axis, use_gs, gs, is_affine, is_pc, eval_mode, init_mode, group, gnumel, reduce) = ctx.cfg
        dx, ds, db = sharded_backward(grad_out, x, scale, shift, qmin, qmax, tmin, tmax, axis, use_gs, gs, is_affine,
                                      is_pc, eval_mode, init_mode, group, gnumel, reduce=reduce)
        return dx, ds, db, None


def lsq_sharded(x, scale, shift, quant_min=0, quant_max=255, type_min=None, type_max=None, axis=1,
                use_grad_scaling=True, grad_scaler=1., is_affine=True, is_perchannel=False,
                eval_mode=False, init_mode=False, group=None, global_numel=None, reduce=True):
    """`torchlsq.functional.lsq` for a tensor whose dim 0 is sharded across the ranks of `group`.

    scale/shift are replicated; their gradients come back already summed over all ranks and equal
    (to the 1e-6 parity budget) the gradients of the unsharded op on the concatenated tensor.
    `global_numel`: the element count of the whole batch for the gradient scaler -- None: local numel x world size (equal
    shards); an int; or `COLLECTIVE` ("collective"): shards may be uneven or empty and no rank knows the total, so the
    count is summed in the same all-reduce (one collective per backward either way).
    `reduce=False`: no collective at all -- scale.grad / shift.grad are the rank's own sums, scaled with the global count
    (equal shards or an int), for a wrapper that reduces gradients itself (DistributedDataParallel).
    Per-channel quantisation along the sharded dim itself (axis 0) needs no collective and is not
    handled here -- use the plain op on each shard."""
    _assert_has_ops()
    if not is_affine:
        assert quant_min <= 0 <= quant_max, 'quantization range must be covered 0 in symmetric quantization'
    assert not (is_perchannel and axis == 0), "axis 0 is the sharded dim: channels are disjoint, use lsq() per shard"
    type_min = quant_min if type_min is None else type_min
    type_max = quant_max if type_max is None else type_max
    if scale.dim() != 1:
        raise RuntimeError("scale should be a 1-D tensor, even in per tensor case(please, avoid torch.Scalar too)")
    if shift.dim() != 1:
        raise RuntimeError("shift should be a 1-D tensor, even in per tensor case(please, avoid torch.Scalar too)")
    if is_perchannel:
        size = max(scale.size(0), shift.size(0))
        scale = scale if scale.size(0) == size else scale.repeat(size)
        shift = shift if shift.size(0) == size else shift.repeat(size)
    cfg = (quant_min, quant_max, type_min, type_max, axis, use_grad_scaling, grad_scaler, is_affine, is_perchannel,
           eval_mode, init_mode, group, global_numel, bool(reduce))
    return _ShardedLSQ.apply(x, scale, shift, cfg)


def all_reduce_minmax(cur_min, cur_max, group=None):
    """Batch min / max over all ranks in ONE collective: [min, -max] packed, all-reduce(MIN).  What the observer-driven
    initialisation of a replicated quantizer needs so that every rank derives the same scale / shift from the whole batch
    (reference quantized/modules/observers.py:446-449 sees the whole batch on its one device).  Returns new tensors."""
    n = cur_min.numel()
    # torch.aminmax (what the reference's observers see, quantized/modules/observers.py:446-449) makes BOTH results NaN when
    # the batch holds a NaN; what MIN does with a NaN is the backend's business (RCCL and gloo differ, and ranks could end up
    # with different parameters exactly when the data is bad).  So a third slot per value carries "this rank saw a NaN" (-1,
    # else 0: NaN-free, the same on every rank after the MIN) and every rank overwrites its result with NaN where any rank
    # raised it -- whatever the backend made of the NaN slots themselves: the aminmax answer on the whole batch, everywhere.
    # (cur_min / cur_max come from aminmax-style reductions: one is NaN exactly when the other is.)
    lo, hi = cur_min.reshape(-1), cur_max.reshape(-1)
    packed = torch.cat([lo, -hi, -(torch.isnan(lo).to(lo.dtype))])
    if _world(group) > 1:
        comm = native_comm(group, packed.device) if (packed.is_cuda and packed.dtype in (torch.float32, torch.float64)) else None
        if comm is not None:
            comm.all_reduce(packed, op=_E.LSQ_COMM_MIN)
        else:
            dist.all_reduce(packed, op=dist.ReduceOp.MIN, group=group)
    poisoned = packed[2 * n:] < 0
    gmin = packed[:n].masked_fill(poisoned, float("nan"))
    gmax = (-packed[n:2 * n]).masked_fill(poisoned, float("nan"))
    return gmin.reshape(cur_min.shape), gmax.reshape(cur_max.shape)
