"""GPU, world_size 2 over gloo on ONE device: torchlsq.distributed on the HIP kernels.

Two processes share cuda:0 (the 1-GPU test box; the RCCL transport itself cannot be exercised there -- RCCL refuses two
ranks on one device -- so the collective runs over gloo, which stages the 16-byte fp64 pair through the host).  What this
pins on the device: the `*_wide` ops (un-rounded fp64 sums, global element count in the gradient scaler), the single
all-reduce, and that the sharded result equals the reference's (the pinned oracle's) on the concatenated tensor: y, dx
bit-exact per shard; d_scale / d_shift within 1e-6 * sum|terms| (north_star's bar), and the GPU's own unsharded op bit for bit
in y / dx.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, per_channel, dtype_name, out_q, shape=(64, 48, 14, 14)):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
    import torchlsq  # noqa: F401
    from torchlsq import extension, synth
    from torchlsq.distributed import lsq_sharded
    from torchlsq.functional import lsq
    extension._assert_has_ops()
    dev = torch.device("cuda:0")
    dtype = getattr(torch, dtype_name)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = {"n": 0}
        real_all_reduce = dist.all_reduce

        def counting_all_reduce(*a, **k):
            calls["n"] += 1
            return real_all_reduce(*a, **k)
        dist.all_reduce = counting_all_reduce

        C = shape[1]
        n = int(np.prod(shape))
        x = synth.normal_like(n, 71, 0.3, 1.0, dtype=dtype, device=dev).view(shape)
        g = synth.normal_like(n, 72, 0.0, 1e-2, dtype=dtype, device=dev).view(shape).abs()     # no cancellation: plain rtol
        if per_channel:
            scale = synth.uniform_like(C, 73, 0.05, 0.3, device=dev)
            shift = synth.normal_like(C, 74, 0.0, 0.1, device=dev)
            kw = dict(quant_min=-8, quant_max=7, type_min=-128, type_max=127, axis=1, is_perchannel=True)
        else:
            scale, shift = torch.tensor([0.03], device=dev), torch.tensor([0.05], device=dev)
            kw = dict(quant_min=0, quant_max=127, type_min=0, type_max=255)
        xf = x.clone().requires_grad_(True)
        sf = scale.clone().requires_grad_(True)
        bf = shift.clone().requires_grad_(True)
        lsq(xf, sf, bf, **kw).backward(g)                       # the unsharded op on the whole batch
        yf = lsq(xf.detach(), scale, shift, **kw)
        h = shape[0] // world
        sl = slice(rank * h, (rank + 1) * h)
        xs = x[sl].clone().requires_grad_(True)
        ss = scale.clone().requires_grad_(True)
        bs = shift.clone().requires_grad_(True)
        ys = lsq_sharded(xs, ss, bs, **kw)
        ys.backward(g[sl])
        torch.cuda.synchronize()
        # the bar is north_star's: against the ORACLE (pinned to the reference CPU csrc) on the CONCATENATED tensor -- fp32 math on
        # the stored values, for 16-bit storage too --, |got - ref| <= 1e-6 * sum|terms| per channel; y and dx bit-exact
        sys.path.insert(0, ROOT)
        from oracle import lsq_oracle as O
        xn, gn = x.float().cpu().numpy(), g.float().cpu().numpy()
        q4 = (kw["quant_min"], kw["quant_max"], kw["type_min"], kw["type_max"])
        if per_channel:
            outer, Cc, inner = O.axis_to_ocl(xn.shape, 1)
            ref = O.bwd_pc(gn, xn, scale.cpu().numpy(), shift.cpu().numpy(), outer, Cc, inner, *q4, True, 1.0, False)
            oy = O.fwd_pc(xn, scale.cpu().numpy(), shift.cpu().numpy(), outer, Cc, inner, *q4)
        else:
            ref = O.bwd_pt(gn, xn, float(scale[0]), float(shift[0]), *q4, True, 1.0, False)
            oy = O.fwd_pt(xn, float(scale[0]), float(shift[0]), *q4)
        err_s = float((np.abs(ss.grad.double().cpu().numpy() - ref.ds_wide) / np.maximum(ref.abs_ds, 1e-300)).max())
        err_b = float((np.abs(bs.grad.double().cpu().numpy() - ref.db_wide) / np.maximum(ref.abs_db, 1e-300)).max())
        oy_t = torch.from_numpy(oy.reshape(shape)).to(dtype)
        odx_t = torch.from_numpy(ref.dx.reshape(shape)).to(dtype)
        ok = (torch.equal(ys.cpu(), oy_t[sl]) and torch.equal(xs.grad.cpu(), odx_t[sl]) and err_s <= 1e-6 and err_b <= 1e-6
              and torch.equal(ys, yf[sl]) and torch.equal(xs.grad, xf.grad[sl]) and calls["n"] == 1)
        out_q.put((rank, bool(ok), calls["n"], err_s, err_b))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("per_channel", [False, True])
@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_sharded_equals_unsharded_on_the_gpu(per_channel, dtype_name):
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, per_channel, dtype_name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok, ncalls, err_s, err_b in res:
        assert ok, "rank %d: sharded != oracle on the whole batch (all_reduce calls %d, err / sum|terms| ds %g db %g)" % (rank, ncalls, err_s, err_b)


def test_config4_per_rank_shard_over_two_ranks():
    """BASELINE config 4 is [1024,1024,14,14] over 8 GPUs: [128,1024,14,14] per rank.  Two ranks with exactly that shard
    (a [256,1024,14,14] global batch, 51 M elements) against the unsharded op on the whole batch: the kernels, grid and
    gradient-scaler count a rank of the 8-GPU job runs, with the collective over gloo."""
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, False, "float32", q, (256, 1024, 14, 14))) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok, ncalls, err_s, err_b in res:
        assert ok, "rank %d: sharded != oracle on the whole batch (all_reduce calls %d, err / sum|terms| ds %g db %g)" % (rank, ncalls, err_s, err_b)
