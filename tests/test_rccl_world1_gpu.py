"""GPU: the collectives of the sharded path executed by RCCL itself (backend "nccl" on ROCm) -- with the ONE rank a 1-GPU box
can give it.  A world of one makes every all-reduce an identity, which is exactly what lets the result be checked: with
torchlsq.distributed told to communicate as if there were peers, the sharded op must equal the plain op on the same tensor.
What this pins that the gloo tests cannot: ProcessGroupNCCL accepts the buffers this code hands it (fp64 SUM of 2C+1 slots,
fp32 MIN of the packed [min, -max]), on the stream the kernels run on, with async_op handles consumed a step later as bench.py
does -- the call sequence the 8-GPU job makes on every rank."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CODE = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "lsqfakequantize-pytorch_amd"))
import torch, torch.distributed as dist
import torchlsq
from torchlsq import distributed as D, synth
from torchlsq.functional import lsq
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
calls = {"n": 0}
real = dist.all_reduce
def counting(*a, **k):
    calls["n"] += 1
    return real(*a, **k)
dist.all_reduce = counting
from torchlsq import _hip_host as _HH
for _name in ("all_reduce", "begin"):          # ... and the library's own communicator (the default route on the GPU)
    def _wrap(_orig):
        def f(self, *a, **k):
            calls["n"] += 1
            return _orig(self, *a, **k)
        return f
    setattr(_HH.HipComm, _name, _wrap(getattr(_HH.HipComm, _name)))
D.assume_peers(True)                    # communicate as if there were peers
_c0 = D.native_comm(None, dev)                                                                          # (a collective call: id broadcast + agreement)
assert _c0 is not None, "the library's RCCL communicator could not be created: %%r" %% (D.LAST_FAILURE,)
# ... and it was CHECKED before anything relies on it: the timed route carried its 128 changing reductions with unfenced events,
# the communicator's stream was picked by lsq_hip_comm_tune (a setup call), and begin itself never had to
_i0 = _c0.info()
assert _c0.checked and _c0.checked["events"].startswith("no system-scope fence") and "all exact" in _c0.checked["route_check"], _c0.checked
assert _i0["side_stream_choice"] >= 2 and _i0["event_system_fence"] == 0 and _i0["reductions_begun"] >= D.ROUTE_CHECK_REDUCTIONS, _i0
assert D.check_timed_route(_c0, dev, 30.0) == ""
calls["n"] = 0
ok = True
for per_channel in (False, True):
    for dtype in (torch.float32, torch.bfloat16):
        shape = (32, 256, 14, 14)
        n = 32 * 256 * 14 * 14
        x = synth.normal_like(n, 3, 0.3, 1.0, dtype=dtype, device=dev).view(shape)
        g = synth.normal_like(n, 4, 0.0, 1e-2, dtype=dtype, device=dev).view(shape).abs()
        if per_channel:
            scale, shift = synth.uniform_like(256, 5, 0.05, 0.3, device=dev), synth.normal_like(256, 6, 0.0, 0.1, device=dev)
            kw = dict(quant_min=-8, quant_max=7, type_min=-128, type_max=127, axis=1, is_perchannel=True)
        else:
            scale, shift = torch.tensor([0.03], device=dev), torch.tensor([0.05], device=dev)
            kw = dict(quant_min=0, quant_max=127, type_min=0, type_max=255)
        xf, sf, bf = x.clone().requires_grad_(True), scale.clone().requires_grad_(True), shift.clone().requires_grad_(True)
        lsq(xf, sf, bf, **kw).backward(g)
        for mode in (D.COLLECTIVE, n):          # the count in the collective / known up front
            xs, ss, bs = x.clone().requires_grad_(True), scale.clone().requires_grad_(True), shift.clone().requires_grad_(True)
            before = calls["n"]
            D.lsq_sharded(xs, ss, bs, global_numel=mode, **kw).backward(g)
            torch.cuda.synchronize()
            if mode == D.COLLECTIVE:
                # the count travels in the collective: the gradient scaler multiplies the fp64 SUM once instead of every fp32 term
                # (include/lsq_hip.h, lsq_hip_sharded_finish) -- up to 2^-24 per term relative to sum|terms|, i.e. inside
                # north_star's 1e-6 * sum|terms|; measured here against |sum| itself, which on these mostly one-signed sums
                # (grad = |.|) is within a factor two of sum|terms|: hence 2e-6, not a looser bar
                sums = torch.allclose(ss.grad, sf.grad, rtol=2e-6, atol=0) and torch.allclose(bs.grad, bf.grad, rtol=2e-6, atol=1e-12)
            else:
                # the count known up front: an identity all-reduce of the same fp64 sums, rounded once -- the plain op's bits
                sums = torch.equal(ss.grad, sf.grad) and torch.equal(bs.grad, bf.grad)
            good = torch.equal(xs.grad, xf.grad) and sums and calls["n"] - before == 1
            ok = ok and good
            print(per_channel, dtype, mode if mode == D.COLLECTIVE else "int", good, calls["n"] - before, flush=True)
# bench.py's pattern: the collective issued async, consumed one step later
pending = None
for step in range(4):
    dx, wide, work = D.sharded_backward(g, x, scale, shift, -8, 7, -128, 127, 1, True, 1.0, True, True, False, False, None, n, async_op=True)
    if pending is not None:
        pending[1].wait()
        _ = pending[0].to(torch.float32)
    pending = (wide, work)
pending[1].wait()
torch.cuda.synchronize()
ok = ok and torch.isfinite(pending[0]).all().item()
# native route: the reduction's first consumer -- the rounding to the parameter type -- ran on the communicator's stream (one host
# call through the C++ binding: lsq_backward_*_sharded; through Python otherwise); after the join it equals the cast of the sums
ok = ok and getattr(pending[1], "deferred", False) and torch.equal(pending[1].rounded, pending[0].to(torch.float32))
from torchlsq import extension as _E2
_E2.set_host_binding("ctypes")
dx_c, wide_c, work_c = D.sharded_backward(g, x, scale, shift, -8, 7, -128, 127, 1, True, 1.0, True, True, False, False, None, n, async_op=True)
work_c.wait()
torch.cuda.synchronize()
ok = ok and torch.equal(wide_c, pending[0]) and torch.equal(work_c.rounded, pending[1].rounded) and torch.equal(dx_c, dx)
_E2.set_host_binding("native")
print("deferred consumer: fused op == python route", ok, flush=True)
dx_j, wide_j, work_j = D.sharded_backward(g, x, scale, shift, -8, 7, -128, 127, 1, True, 1.0, True, True, False, False, None, n, async_op=True)
D.join(None, dev)                       # the module-level form of work.wait(): everything begun on the communicator so far
torch.cuda.synchronize()
ok = ok and torch.equal(work_j.rounded, pending[1].rounded)
# the observer statistics' packed MIN all-reduce
lo, hi = torch.tensor([-1.5, 0.25], device=dev), torch.tensor([2.0, 0.75], device=dev)
a, b = D.all_reduce_minmax(lo, hi, None)
torch.cuda.synchronize()
ok = ok and torch.equal(a, lo) and torch.equal(b, hi)
# the drop-in module with rank sync, told it has a peer: observer-driven init batches (fused statistics -> packed MIN all-reduce
# -> one-launch observer tail) and LSQ steps (count in the collective) must reproduce the plain module call for call
from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
from torchlsq.quantized import LSQFakeQuantizer
from torchlsq.quantized.modules import observers as OBS
OBS._dist_world = lambda group: 2
for obs_cls, extra in ((MovingAverageMinMaxObserver, {}), (MovingAveragePerChannelMinMaxObserver, dict(qscheme=torch.per_channel_affine, ch_axis=1))):
    a = LSQFakeQuantizer(obs_cls, "activation", init_batches=2, sync=True, **extra).train()
    p = LSQFakeQuantizer(obs_cls, "activation", init_batches=2, **extra).train()
    for i in range(6):
        xi = synth.normal_like(8 * 16 * 6 * 6, 50 + i, 0.8, 1.0, device=dev).view(8, 16, 6, 6)
        wi = synth.normal_like(8 * 16 * 6 * 6, 70 + i, 0.0, 1.0, device=dev).view(8, 16, 6, 6)
        xa, xp = xi.clone().requires_grad_(True), xi.clone().requires_grad_(True)
        before = calls["n"]
        ya, yp = a(xa), p(xp)
        if i == 0:
            a.to(dev); p.to(dev)
        if ya.requires_grad:
            for m in (a, p):
                m.scale.grad = None; m.shift.grad = None
            (ya * wi).sum().backward(); (yp * wi).sum().backward()
        torch.cuda.synchronize()
        good = torch.equal(ya.detach(), yp.detach()) and torch.equal(a.scale, p.scale) and torch.equal(a.shift, p.shift)
        if a.scale.grad is not None and p.scale.grad is not None:
            # (the synchronised module takes the counted route above -- scaler on the sum -- and these activation gradients are
            #  mixed-sign sums: |sum| is ~10 x smaller than sum|terms|, so 1e-6 * sum|terms| reads as 2e-5 of the sum; the arithmetic
            #  bar itself is held against the oracle in tests/test_module_sync_gpu.py / test_sharded_gpu.py)
            good = good and torch.allclose(a.scale.grad, p.scale.grad, rtol=2e-5, atol=1e-12) and torch.allclose(a.shift.grad, p.shift.grad, rtol=2e-5, atol=1e-10)
        ok = ok and good
        print("module", obs_cls.__name__, "call", i, good, "collectives", calls["n"] - before, flush=True)
# ---- the library's own communicator (include/lsq_hip.h, lsq_hip_comm_*): everything above ran over it -- the default route
# for GPU tensors over an RCCL group -- which the counter of torch.distributed.all_reduce calls cannot see: say so explicitly
comm = D.native_comm(None, dev, create=False)
ok = ok and comm is not None
print("native communicator", comm.info() if comm is not None else None, flush=True)
if comm is not None:
    from torchlsq import extension as E
    info = comm.info()
    ok = ok and info["rank"] == 0 and info["nranks"] == 1 and info["rccl_version"] > 0
    # in place, out of place, the three reductions, both element types, on a side stream of the caller's
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for dt in (torch.float64, torch.float32):
            src = torch.arange(1, 9, dtype=dt, device=dev) * 0.37
            for op in (E.LSQ_COMM_SUM, E.LSQ_COMM_MIN, E.LSQ_COMM_MAX):
                a = src.clone()
                comm.all_reduce(a, op=op)
                out = torch.zeros_like(src)
                comm.all_reduce(src, op=op, out=out)
                t1 = comm.begin(src, op=op, out=out.zero_())
                t2 = comm.begin(a, op=op)
                comm.end(t1); comm.end(t2)
                st.synchronize()
                ok = ok and torch.equal(a, src) and torch.equal(out, src)
    # the overlapped form inside a HIP-graph capture: every begin ended before the capture ends
    gbuf = torch.full((3,), 2.5, dtype=torch.float64, device=dev)
    gout = torch.zeros_like(gbuf)
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            gbuf.mul_(2.0)
            tk = comm.begin(gbuf, out=gout)
            gbuf2 = gbuf + 1.0                  # work that overlaps the reduction
            comm.end(tk)
            res = gout + gbuf2
        gr.replay(); gr.replay()
    torch.cuda.synchronize()
    ok = ok and torch.equal(gbuf, torch.full((3,), 10.0, dtype=torch.float64, device=dev)) and torch.equal(res, gbuf + gbuf + 1.0)
    print("native comm: reductions + capture", ok, flush=True)
    # the same sharded calls over torch.distributed instead (TORCHLSQ_COLLECTIVE=c10d / set_native_collective(False)): counted
    D.set_native_collective(False)
    before = calls["n"]
    xs, ss, bs = x.clone().requires_grad_(True), scale.clone().requires_grad_(True), shift.clone().requires_grad_(True)
    D.lsq_sharded(xs, ss, bs, global_numel=D.COLLECTIVE, quant_min=-8, quant_max=7, type_min=-128, type_max=127, axis=1, is_perchannel=True).backward(g)
    torch.cuda.synchronize()
    ok = ok and calls["n"] - before == 1
    D.set_native_collective(True)
    # bad arguments are rejected with a message, not executed
    try:
        comm.all_reduce(torch.zeros(4, dtype=torch.int32, device=dev))
        ok = False
    except RuntimeError:
        pass
D.destroy_native_comms()
dist.destroy_process_group()
print("RESULT", "ok" if ok else "FAILED", flush=True)
'''


def test_sharded_path_over_rccl_world_of_one():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CODE % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "RESULT ok" in r.stdout, r.stdout[-3000:]


CHECK_CODE = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "lsqfakequantize-pytorch_amd"))
import torch, torch.distributed as dist
import torchlsq
from torchlsq import distributed as D, synth
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
D.assume_peers(True)
comm = D.native_comm(None, dev)
assert comm is None, "a communicator whose first reduction did not finish in time must not be kept"
assert D.native_comm(None, dev, create=False) is None
# ... and the sharded backward goes through torch.distributed, same answer as the plain op
n = 8 * 64 * 14 * 14
x = synth.normal_like(n, 3, 0.3, 1.0, device=dev).view(8, 64, 14, 14)
g = synth.normal_like(n, 4, 0.0, 1e-2, device=dev).view(8, 64, 14, 14)
s, b = torch.tensor([0.03], device=dev), torch.tensor([0.05], device=dev)
dx, wide, work = D.sharded_backward(g, x, s, b, 0, 127, 0, 255, global_numel=n, async_op=True)
assert work is not None and not getattr(work, "deferred", False), "the native route must be off"
work.wait()
dx1, ds1, db1 = D.sharded_backward(g, x, s, b, 0, 127, 0, 255, global_numel=n, reduce=False)
assert torch.equal(dx, dx1) and torch.equal(wide[0].to(torch.float32).reshape(-1), ds1.reshape(-1))
print("FALLBACK_OK", flush=True)
os._exit(0)         # (the communicator that "hung" was left alone on purpose: no teardown)
'''


def test_a_communicator_that_fails_its_first_reduction_is_not_kept():
    """native_comm() checks the communicator before anything relies on it (the ranks add up rank + 1 under a deadline and
    agree); here the deadline is negative -- "it never finished" -- and the path must end on torch.distributed"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", TORCHLSQ_COMM_CHECK_S="-1")
    r = subprocess.run([sys.executable, "-c", CHECK_CODE % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "FALLBACK_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


FENCED_CODE = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "lsqfakequantize-pytorch_amd"))
import torch, torch.distributed as dist
import torchlsq
from torchlsq import distributed as D, synth
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
D.assume_peers(True)
D._TEST_HOOKS["unfenced_fails"] = True
comm = D.native_comm(None, dev)
assert comm is not None, D.LAST_FAILURE
info = comm.info()
assert info["event_system_fence"] == 1 and comm.checked["events"].startswith("system-fenced"), (info, comm.checked)
# the sharded step over the re-configured communicator: same answer as the plain op
n = 8 * 64 * 14 * 14
x = synth.normal_like(n, 3, 0.3, 1.0, device=dev).view(8, 64, 14, 14)
g = synth.normal_like(n, 4, 0.0, 1e-2, device=dev).view(8, 64, 14, 14)
s, b = torch.tensor([0.03], device=dev), torch.tensor([0.05], device=dev)
dx, wide, work = D.sharded_backward(g, x, s, b, 0, 127, 0, 255, global_numel=n, async_op=True)
assert getattr(work, "deferred", False)
work.wait()
dx1, ds1, db1 = D.sharded_backward(g, x, s, b, 0, 127, 0, 255, global_numel=n, reduce=False)
torch.cuda.synchronize()
assert torch.equal(dx, dx1) and torch.equal(work.rounded[0].reshape(-1), ds1.reshape(-1)) and torch.equal(work.rounded[1].reshape(-1), db1.reshape(-1))
# configure back and forth is a setup call that leaves a working communicator
comm.configure(event_system_fence=False)
assert comm.info()["event_system_fence"] == 0 and D.check_timed_route(comm, dev, 30.0) == ""
# a second communicator of the process takes the parked stream of a destroyed one over (no stream left behind per communicator)
side0 = comm.side_stream().cuda_stream
D.destroy_native_comms()
D._TEST_HOOKS.clear()
comm2 = D.native_comm(None, dev)
assert comm2 is not None and comm2 is not comm
import ctypes
from torchlsq import _abi
cands_first = comm2.side_stream().cuda_stream
print("parked stream reused or re-picked:", side0, cands_first, flush=True)
D.destroy_native_comms()
dist.destroy_process_group()
print("FENCED_OK", flush=True)
'''


def test_the_system_fenced_fallback_of_the_route_check():
    """native_comm step 5: when the unfenced events fail the timed route's self-check on any rank, every rank re-configures the
    communicator with system-fenced events, checks again and keeps it -- here forced by a test hook, in an RCCL world of one"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", FENCED_CODE % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "FENCED_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
