"""CPU, world_size 2 over gloo: the batch-sharded path (torchlsq.distributed) is correct by construction.

Each rank owns half of dim 0; with the oracle plugged in as the CPU kernel the sharded op's output and
gradients must equal the unsharded op on the concatenated tensor: y/dx bit-exact per shard, d_scale /
d_shift equal after the ONE all-reduce (the global element count feeds the gradient scaler).
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, per_channel, out_q, bounds=None, collective=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
    sys.path.insert(0, ROOT)
    import conftest
    conftest.install_oracle_cpu_backend()
    from torchlsq import synth
    from torchlsq.distributed import lsq_sharded
    from torchlsq.functional import lsq
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = {"n": 0}
        real_all_reduce = dist.all_reduce

        def counting_all_reduce(*a, **k):
            calls["n"] += 1
            return real_all_reduce(*a, **k)
        dist.all_reduce = counting_all_reduce

        shape = (6, 8, 5, 5) if bounds is None else (bounds[-1], 8, 5, 5)
        n = int(np.prod(shape))
        x = synth.normal_like(n, 71, 0.3, 1.0).view(shape)
        g = synth.normal_like(n, 72, 0.0, 1e-2).view(shape)
        if per_channel:
            scale = synth.uniform_like(8, 73, 0.05, 0.3)
            shift = synth.normal_like(8, 74, 0.0, 0.1)
            kw = dict(quant_min=-8, quant_max=7, type_min=-128, type_max=127, axis=1, is_perchannel=True)
        else:
            scale, shift = torch.tensor([0.03]), torch.tensor([0.05])
            kw = dict(quant_min=0, quant_max=127, type_min=0, type_max=255)
        # unsharded reference on the whole batch
        xf = x.clone().requires_grad_(True)
        sf = scale.clone().requires_grad_(True)
        bf = shift.clone().requires_grad_(True)
        yf = lsq(xf, sf, bf, **kw)
        yf.backward(g)
        # this rank's shard
        if bounds is None:       # equal shards: the global element count is local numel x world size (the default)
            h = shape[0] // world
            sl = slice(rank * h, (rank + 1) * h)
            extra = {}
        else:                    # UNEVEN shards (one of them may be empty): the caller states the global element count
            sl = slice(bounds[rank], bounds[rank + 1])
            extra = dict(global_numel="collective" if collective else n)     # COLLECTIVE: the count rides in the all-reduce
        xs = x[sl].clone().requires_grad_(True)
        ss = scale.clone().requires_grad_(True)
        bs = shift.clone().requires_grad_(True)
        ys = lsq_sharded(xs, ss, bs, **kw, **extra)
        ys.backward(g[sl])
        ok = (torch.equal(ys, yf[sl]) and torch.equal(xs.grad, xf.grad[sl])
              and torch.allclose(ss.grad, sf.grad, rtol=1e-6, atol=0) and torch.allclose(bs.grad, bf.grad, rtol=1e-6, atol=1e-12)
              and calls["n"] == 1)
        # without the global-numel scaler a plain per-shard backward would NOT match: guard the contract
        out_q.put((rank, bool(ok), calls["n"], float((ss.grad - sf.grad).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("per_channel", [False, True])
def test_sharded_equals_unsharded_gloo_world2(per_channel):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, per_channel, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ncalls, err in res:
        assert ok, "rank %d: sharded != unsharded (all_reduce calls %d, max |ds err| %g)" % (rank, ncalls, err)


@pytest.mark.parametrize("per_channel", [False, True])
def test_uneven_shards_explicit_global_numel_gloo_world4(per_channel):
    """world 4, shards of 5 / 1 / 3 / 2 rows of an 11-row batch: with `global_numel` given, the sharded op still equals the
    unsharded one on the concatenated tensor (one all-reduce per backward on every rank)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    bounds = (0, 5, 6, 9, 11)
    procs = [ctx.Process(target=_worker, args=(r, 4, port, per_channel, q, bounds)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1, 2, 3]
    for rank, ok, ncalls, err in res:
        assert ok, "rank %d: sharded != unsharded (all_reduce calls %d, max |ds err| %g)" % (rank, ncalls, err)


@pytest.mark.parametrize("per_channel", [False, True])
def test_uneven_shards_count_in_the_collective_gloo_world4(per_channel):
    """the same 5 / 1 / 3 / 2 rows with global_numel=COLLECTIVE: no rank is told the total; the element count is summed in the
    ONE all-reduce and the scaler derived from it (lsq_cpu_sharded_finish) -- still one collective per backward, same result"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    bounds = (0, 5, 6, 9, 11)
    procs = [ctx.Process(target=_worker, args=(r, 4, port, per_channel, q, bounds, True)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ncalls, err in res:
        assert ok, "rank %d: sharded != unsharded (all_reduce calls %d, max |ds err| %g)" % (rank, ncalls, err)
