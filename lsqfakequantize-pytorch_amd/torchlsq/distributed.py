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
import threading
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
# an RCCL ("nccl") process group -- see native_comm for the protocol: every step of it is agreed between the ranks over
# torch.distributed, and any failure leaves ALL of them on torch.distributed.
# TORCHLSQ_COLLECTIVE=c10d switches the native route off; gloo groups and CPU tensors never take it.
_COMMS = {}                 # (group key, device index) -> (HipComm or None: this group stays on torch.distributed, its process-group object)
_NATIVE_COLLECTIVE = [os.environ.get("TORCHLSQ_COLLECTIVE", "native").lower() != "c10d"]
ROUTE_CHECK_REDUCTIONS = 128    # reductions of the timed route's self-check (native_comm step 5)


def set_native_collective(on):
    """True: GPU tensors over an RCCL group reduce through the library's own communicator (the default); False: always
    torch.distributed.all_reduce.  Communicators already created stay alive but are not used while it is off."""
    _NATIVE_COLLECTIVE[0] = bool(on)


def _group_key(group):
    return 0 if group is None else id(group)


def _process_group(group):
    """the process-group OBJECT behind `group` (None = the default group): what a cache entry is tied to"""
    return group if group is not None else dist.distributed_c10d._get_default_group()


def _agree(values, device, group):
    """MIN over the ranks of a few small ints, through torch.distributed (its own communicator and stream)"""
    flag = torch.tensor([int(v) for v in values], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return [int(v) for v in flag.tolist()]


def _wait_event(ev, limit):
    """poll a recorded event under a host-side deadline; False = it did not finish (a negative limit, tests: 'it never did')"""
    deadline = time.monotonic() + limit
    while not ev.query() and time.monotonic() < deadline:
        time.sleep(0.002)
    return limit >= 0 and ev.query()


def _create_under_deadline(uid, rank, ws, device, limit):
    """HipComm(...) -- ncclCommInitRank, a collective -- on a helper thread: the caller gets (comm, None), (None, error text) or
    (None, None) when the bootstrap did not return within `limit` seconds (the thread is a daemon and is left to it)."""
    box = {}

    def work():
        try:
            box["comm"] = _E.HipComm(uid, rank, ws, device)
        except Exception as e:           # noqa: BLE001 -- anything: the caller falls back
            box["error"] = "%s: %s" % (type(e).__name__, e)

    th = threading.Thread(target=work, daemon=True, name="lsq-comm-create")
    th.start()
    th.join(limit)
    if th.is_alive():
        return None, None
    return box.get("comm"), box.get("error", "no communicator")


def check_timed_route(comm, device, limit, n=ROUTE_CHECK_REDUCTIONS):
    """The route sharded_backward(async_op=True) takes, exercised BEFORE anything relies on it: `n` reductions whose value
    changes every time, each produced on the current stream, reduced with begin on the communicator's stream, consumed there
    (a copy behind the reduction, like the rounding of the real route) and, after a join every fourth reduction, consumed
    again on the current stream; four buffers in rotation, each rewritten right after the join that covers its last reader.
    A missing ordering anywhere -- producer -> reduction (the `ready` event), reduction -> consumer on the communicator's stream
    (stream order), reduction -> consumer on the current stream (the join's event) -- shows up as a stale or torn value.
    Returns "" (every one of the 2 n consumed values is the exact sum), "hung" or what was wrong."""
    ws, rank = comm.nranks, comm.rank
    cur = torch.cuda.current_stream(device)
    side = comm.side_stream()
    bufs = [torch.zeros(3, dtype=torch.float64, device=device) for _ in range(4)]
    seen_side = torch.zeros(n, 3, dtype=torch.float64, device=device)
    seen_cur = torch.zeros(n, 3, dtype=torch.float64, device=device)
    base = torch.arange(n, dtype=torch.float64, device=device) * 1000.0 + float(rank + 1)
    torch.cuda.synchronize(device)
    for k in range(n):
        b = bufs[k % 4]
        b.copy_(base[k].expand(3))                 # the producer, on the current stream: k * 1000 + rank + 1
        comm.begin(b)
        with torch.cuda.stream(side):
            seen_side[k].copy_(b)                  # the consumer behind the reduction, on the communicator's stream
        if k % 4 == 3 or k == n - 1:
            comm.join()
            lo = k - (k % 4)
            for j in range(lo, k + 1):
                seen_cur[j].copy_(bufs[j % 4])     # ... and one on the current stream, after the join
    for t in bufs + [seen_side, seen_cur, base]:
        t.record_stream(side)
    ev = torch.cuda.Event()
    ev.record(cur)
    if not _wait_event(ev, limit):
        return "hung"
    want = (torch.arange(n, dtype=torch.float64) * 1000.0 * ws + ws * (ws + 1) / 2.0).unsqueeze(1).expand(n, 3)
    for name, got in (("communicator's stream", seen_side.cpu()), ("current stream after join", seen_cur.cpu())):
        bad = (got != want).any(dim=1).nonzero().reshape(-1)
        if bad.numel():
            k = int(bad[0])
            return "%d of %d reductions wrong when consumed on the %s; first: #%d got %r, expected %r" % (
                bad.numel(), n, name, k, got[k].tolist(), float(want[k, 0]))
    return ""


def nothung_of(hung):
    """the 'nobody hung' slot of an agreement: 0, or -1 where this rank saw a reduction / bootstrap that did not finish (MIN)"""
    return -int(bool(hung))


def native_comm(group, device, create=True):
    """The HipComm of (group, device), created on first use -- a COLLECTIVE call then: every rank of the group must make its
    first sharded call at the same point, which data-parallel training does by construction.  None = torch.distributed.
    create=False only looks: it never creates and never decides anything for later calls.

    A communicator is only kept if it WORKS, and no rank enters a step another rank skips -- every decision is a MIN over the
    ranks through torch.distributed (host-side deadline per step: TORCHLSQ_COMM_CHECK_S, 60 s):
      1. rank 0 makes the 128-byte id, every rank checks that the library can reach RCCL; the id is broadcast;
      2. AGREE: all ranks can go on -- otherwise nobody calls ncclCommInitRank (a rank waiting in RCCL's bootstrap for one that
         never comes would wait forever);
      3. lsq_hip_comm_create on a helper thread, under the deadline; lsq_hip_comm_tune on the current stream (the
         communicator's stream is picked against the stream the steps run on; setup call: allocates, synchronises);
      4. the ranks add up rank + 1 through it in stream order, on a stream of its own, under the deadline;
      5. the route of sharded_backward(async_op=True) -- begin on the communicator's stream, a consumer behind it there, one
         join -- carries ROUTE_CHECK_REDUCTIONS reductions of changing values (check_timed_route) with the events recorded
         WITHOUT the system-scope fence; AGREE; if any rank saw a wrong value: lsq_hip_comm_configure(event_system_fence = 1)
         on every rank and the same check again;
      6. AGREE: kept only if every rank passed 4 and 5.  `comm.checked` records what was found (bench.py prints it)."""
    if not _NATIVE_COLLECTIVE[0] or not (dist.is_available() and dist.is_initialized()):
        return None
    key = (_group_key(group), device.index)
    pg = _process_group(group)
    hit = _COMMS.get(key)
    if hit is not None:
        # a hit must still belong to THIS world: dist.destroy_process_group() + a new init hands out a new process-group object
        # (and the entry's own reference keeps an explicit group's id() from being recycled while it is cached) -- one identity
        # test on the per-step path; a communicator destroyed behind the cache's back is dropped too
        comm, owner = hit
        if owner is pg and (comm is None or comm.handle):
            return comm
        del _COMMS[key]
    if not create:
        return None
    if dist.get_backend(group) != "nccl":
        _COMMS[key] = (None, pg)
        return None
    rank, ws = dist.get_rank(group), dist.get_world_size(group)
    limit = float(os.environ.get("TORCHLSQ_COMM_CHECK_S", "60"))       # (tests: a negative limit = "it never finished")
    t0 = time.monotonic()
    checked = {"world": ws}
    ok = 1
    uid = torch.zeros(_E.LSQ_COMM_ID_BYTES, dtype=torch.uint8, device=device)
    try:                                                        # 1.
        mine = _E.HipComm.unique_id()                           # (every rank: it is also the "RCCL resolves here" probe)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(mine), dtype=torch.uint8))
    except Exception as e:       # noqa: BLE001 -- no RCCL behind the library: still take part in the collectives below
        ok, checked["why"] = 0, "RCCL not reachable from the library: %s" % e
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(uid, src=src, group=group)
    if _agree([ok], device, group)[0] != 1:                     # 2.
        _COMMS[key] = (None, pg)
        LAST_FAILURE.clear()
        LAST_FAILURE.update(checked, hung=False, why=checked.get("why", "another rank cannot reach RCCL"))
        return None
    comm, err = _create_under_deadline(bytes(uid.cpu().numpy().tobytes()), rank, ws, device, limit if limit > 0 else 60.0)   # 3.
    hung = 0
    if comm is None:
        ok, hung = 0, int(err is None)
        checked["why"] = err or "ncclCommInitRank did not return in time"
    ok, nothung = _agree([ok, nothung_of(hung)], device, group)      # (a rank without peers must not start a reduction)
    hung = -nothung
    unfenced_ok = 1
    if ok == 1:
        try:
            comm.tune()
            pre = torch.cuda.Stream(device=device)              # 4. (a reduction that never ends must not sit in the caller's stream)
            with torch.cuda.stream(pre):
                probe = torch.full((2,), float(rank + 1), dtype=torch.float64, device=device)
                comm.all_reduce(probe)
                ev = torch.cuda.Event()
                ev.record(pre)
            if not _wait_event(ev, limit):
                ok, hung, checked["why"] = 0, 1, "the first reduction (16 bytes, stream order) did not finish in time"
            elif float(probe[0].item()) != ws * (ws + 1) / 2.0:
                ok, checked["why"] = 0, "the first reduction gave %r, expected %r" % (float(probe[0].item()), ws * (ws + 1) / 2.0)
            else:
                why = check_timed_route(comm, device, limit)    # 5., unfenced events
                if not why and _TEST_HOOKS.get("unfenced_fails"):
                    why = "forced by a test hook"
                if why == "hung":
                    ok, hung, checked["why"] = 0, 1, "the timed route's self-check did not finish in time"
                elif why:
                    unfenced_ok, checked["unfenced"] = 0, why
        except Exception as e:       # noqa: BLE001
            ok, checked["why"] = 0, "%s: %s" % (type(e).__name__, e)
        ok, nothung, unfenced_ok = _agree([ok, nothung_of(hung), unfenced_ok], device, group)
        hung = -nothung
    if ok == 1 and unfenced_ok != 1:
        fenced_ok = 1
        try:
            comm.configure(event_system_fence=True)
            why = check_timed_route(comm, device, limit)
            if why:
                fenced_ok, hung, checked["why"] = 0, int(why == "hung"), "with system-fenced events too: " + why
        except Exception as e:       # noqa: BLE001
            fenced_ok, checked["why"] = 0, "%s: %s" % (type(e).__name__, e)
        ok, nothung = _agree([fenced_ok, nothung_of(hung)], device, group)
        hung = -nothung
    if ok != 1:                                                 # 6.
        if comm is not None and not hung:       # (hung anywhere: left alone, a teardown would wait for it too)
            try:
                comm.destroy()
            except Exception:        # noqa: BLE001
                pass
        LAST_FAILURE.clear()
        LAST_FAILURE.update(checked, hung=bool(hung), why=checked.get("why", "another rank failed"))
        comm = None
    else:
        checked.update(route_check="%d reductions of changing values through begin / side stream / join: all exact" % ROUTE_CHECK_REDUCTIONS,
                       events="system-fenced (the unfenced form failed the check on some rank)" if unfenced_ok != 1 else
                              "no system-scope fence (checked)", seconds=round(time.monotonic() - t0, 3))
        comm.checked = checked
    _COMMS[key] = (comm, pg)
    return comm


_TEST_HOOKS = {}            # tests only: {"unfenced_fails": True} makes step 5 take the system-fenced branch
LAST_FAILURE = {}           # why the last native_comm creation fell back (records; empty = none did)


def destroy_native_comms():
    """tear the library's communicators down -- REQUIRED before dist.destroy_process_group() when the native route was used
    (collective per communicator)"""
    for key, (comm, _owner) in list(_COMMS.items()):
        if comm is not None and comm.handle:
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
