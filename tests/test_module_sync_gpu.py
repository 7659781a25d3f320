"""GPU, gloo on ONE device: `LSQFakeQuantizer(sync=True)` on the HIP kernels.

The ranks of tests/test_module_sync_cpu.py share cuda:0 of the 1-GPU test box (RCCL refuses two ranks on one device, so the
packed collectives run over gloo).  What this pins on the device: the fused observer step with the packed [min, -max]
all-reduce between its two launches (lsq_hip_minmax_* -> collective -> lsq_hip_observer_update), the sharded backward with
the element count in the collective (unscaled terms -> lsq_hip_sharded_finish), empty shards, and that every rank lands on
the REFERENCE module's whole-batch trace.
"""
import pytest
import torch

import sync_workers
from test_module_sync_cpu import _run

pytestmark = pytest.mark.gpu


def test_synced_module_replays_the_reference_traces_on_the_gpu_world2():
    assert torch.cuda.is_available()
    _run(sync_workers.replay, 2, False, "cuda:0", timeout=600)


def test_synced_module_replays_the_reference_traces_on_the_gpu_world4_uneven():
    assert torch.cuda.is_available()
    _run(sync_workers.replay, 4, True, "cuda:0", timeout=900)


def test_a_nan_in_one_shard_on_the_gpu():
    """the fused observer step (lsq_hip_minmax_* -> NaN-free packed MIN all-reduce -> lsq_hip_observer_update) with a NaN in one
    rank's shard: the reference module's whole-batch answer on every rank"""
    assert torch.cuda.is_available()
    _run(sync_workers.nan_sync, 2, "cuda:0", timeout=600)


@pytest.mark.parametrize("grads", ["mean", "ddp"])
def test_ddp_on_the_gpu_replicas_identical(grads):
    assert torch.cuda.is_available()
    _run(sync_workers.ddp_train, 2, "cuda:0", grads, timeout=600)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
@pytest.mark.parametrize("per_channel", [False, True])
def test_sharded_finish_equals_the_host_chain(dtype, per_channel):
    """lsq_hip_sharded_finish: the scaler the kernel derives from the device-resident count is the reference's chain
    (lsq_hip_grad_scaler, host), and ds / db are the fp64 sums times it, rounded once; the CPU twin gives the same bits."""
    from torchlsq import extension as E
    C = 37 if per_channel else 1
    g = torch.Generator().manual_seed(5)
    for count, qmax, use_gs, gs in ((25690112, 127, True, 1.0), (3, 7, True, 0.5), (205520896 * 8, 255, True, 2.0), (1000, 15, False, 0.25),
                                    (0, 127, True, 1.0)):
        packed = torch.cat([torch.randn(2 * C, generator=g, dtype=torch.float64) * 1e3, torch.tensor([float(count)], dtype=torch.float64)])
        ds, db = E.hip_sharded_finish(packed.cuda(), C, per_channel, dtype, qmax, use_gs, gs)
        ds_c, db_c = E.cpu_sharded_finish(packed.clone(), C, per_channel, dtype, qmax, use_gs, gs)
        from torchlsq import _abi
        code = _abi._DTYPE_CODE[dtype]
        s = E.library().lsq_hip_grad_scaler(code, 1 if per_channel else 0, count, qmax, C, 1 if use_gs else 0, gs) if count else 0.0
        pd = torch.float64 if dtype == torch.float64 else torch.float32
        want_ds, want_db = (packed[:C] * s).to(pd), (packed[C:2 * C] * s).to(pd)
        assert torch.equal(ds.cpu(), want_ds) and torch.equal(db.cpu(), want_db), (count, qmax)
        assert torch.equal(ds_c, want_ds) and torch.equal(db_c, want_db)
