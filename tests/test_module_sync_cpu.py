"""CPU, gloo: `LSQFakeQuantizer(sync=True)` -- one quantizer over a batch that is sharded across ranks.

The reference module (quantized/modules/observers.py:424-462) sees the whole batch on one device.  With rank sync every rank
feeds its shard and must end up exactly where the reference's trace of the WHOLE batch does: identical scale / shift on all
ranks after every call (observer-driven init, 'learnable' init and LSQ steps), gradients within the parity budget, one
collective per observer step and one per backward.  World 2 (equal shards) and world 4 with UNEVEN shards, one of them empty.
Also: a small QAT model under DistributedDataParallel (`torchlsq.quantized.prepare_ddp`) against single-process training.
"""
import socket

import pytest
import torch.multiprocessing as mp

import sync_workers


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(target, world, *args, timeout=300):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    problems = [msg for _, msgs in sorted(res) for msg in msgs]
    assert not problems, "\n".join(problems[:12])


def test_shard_bounds_helper():
    assert sync_workers.shard_bounds(4, 2, False) == [0, 2, 4]
    assert sync_workers.shard_bounds(4, 4, True) == [0, 2, 2, 3, 4]          # an empty shard
    assert sync_workers.shard_bounds(3, 4, True) == [0, 2, 2, 2, 3]
    assert sync_workers.shard_bounds(16, 4, True)[-1] == 16


def test_synced_module_replays_the_reference_traces_world2():
    _run(sync_workers.replay, 2, False, "cpu")


def test_synced_module_replays_the_reference_traces_world4_uneven_shards():
    _run(sync_workers.replay, 4, True, "cpu", timeout=420)


def test_negative_control_without_sync_the_replicas_leave_the_reference_trace():
    """the same replay with sync off must FAIL its checks (rank-local statistics and scalers): the test above has teeth"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=sync_workers.replay, args=(r, 2, port, False, "cpu", q, ("act_observer_pt", "act_learnable_pt"), False))
             for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    problems = [msg for _, msgs in res for msg in msgs]
    assert any("scale/shift differ from the reference trace" in m for m in problems)
    assert any("grad" in m for m in problems)


def test_a_nan_in_one_shard_gives_every_rank_the_reference_modules_answer():
    """torch.aminmax semantics through the rank-synchronised observer (per tensor and per channel): a NaN in rank 1's shard
    only, both ranks land on the reference module's whole-batch trace, replicas bit-identical"""
    _run(sync_workers.nan_sync, 2, "cpu")


@pytest.mark.parametrize("grads", ["mean", "ddp"])
def test_ddp_replicas_stay_identical_and_follow_single_process_training(grads):
    """grads='mean': the quantizers all-reduce their own gradient sums (count in the collective); 'ddp': no collective of
    their own -- DDP's bucketed all-reduce averages scale.grad / shift.grad like any other gradient"""
    _run(sync_workers.ddp_train, 2, "cpu", grads)


def test_sync_is_a_noop_outside_a_job_and_on_weights():
    import torch
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver as Obs, HistogramObserver
    from torchlsq.quantized import LSQFakeQuantizer as Q
    a, b = Q(Obs, "activation", init_batches=1, sync=True), Q(Obs, "activation", init_batches=1)
    x = torch.rand(4, 8) + 0.5
    for _ in range(3):
        ya, yb = a(x), b(x)
    assert torch.equal(ya, yb) and a._sync_world() == 1
    w = Q(None, "weight", dtype=torch.qint8, qscheme=torch.per_tensor_symmetric, init_mode="learnable", sync=True)
    assert w._sync_world() == 1
    with pytest.raises(AssertionError, match="MinMax family"):
        Q(HistogramObserver, "activation", sync=True)
    with pytest.raises(AssertionError, match="nothing to synchronise"):
        Q(None, "activation", qscheme=torch.per_channel_affine, ch_axis=0, init_mode="learnable", sync=True)
    import copy
    import pickle
    c = copy.deepcopy(a)
    assert c._sync and c._group_ref is a._group_ref
    assert pickle.loads(pickle.dumps(a._group_ref)).group is None


def test_prepare_ddp_lists_what_ddp_need_not_touch(oracle_cpu_backend):
    import torch
    from torch.ao.quantization import QConfig
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torchlsq.quantized import LSQFakeQuantizer, prepare_ddp
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.ReLU())
    model.qconfig = QConfig(activation=LSQFakeQuantizer.with_args(observer=MovingAverageMinMaxObserver, otype="activation"),
                            weight=LSQFakeQuantizer.with_args(observer=MovingAveragePerChannelMinMaxObserver, otype="weight", dtype=torch.qint8,
                                                              qscheme=torch.per_channel_symmetric))
    torch.ao.quantization.prepare_qat(model.train(), inplace=True)
    model(torch.randn(2, 3, 8, 8))
    out = prepare_ddp(model)
    assert out is model
    ig = set(model._ddp_params_and_buffers_to_ignore)
    names = dict(model.named_parameters())
    names.update(dict(model.named_buffers()))
    assert ig <= set(names), sorted(ig - set(names))                      # every entry is a real parameter / buffer name
    act = [n for n, m in model.named_modules() if isinstance(m, LSQFakeQuantizer) and m.dtype == torch.quint8]
    wgt = [n for n, m in model.named_modules() if isinstance(m, LSQFakeQuantizer) and m.dtype == torch.qint8]
    assert act and wgt
    for n in act:
        assert {n + ".scale", n + ".shift", n + ".current_batch", n + ".activation_post_process.min_val"} <= ig
        assert dict(model.named_modules())[n]._sync and dict(model.named_modules())[n]._sync_grads == "mean"
    for n in wgt:            # weights: replicated input, DDP averages their gradients like any parameter's; only the flags are skipped
        assert n + ".scale" not in ig and n + ".fake_quant_enabled" in ig and not dict(model.named_modules())[n]._sync


def test_prepared_model_with_sync_survives_save_load_and_deepcopy(oracle_cpu_backend):
    """torch.save(model) / torch.load and copy.deepcopy of a QAT model whose activation quantizers are rank-synchronised: the
    process group is not part of the pickle (it comes back as the default group) and is shared by a deep copy"""
    import copy
    import io
    import torch
    from torch.ao.quantization import QConfig, prepare_qat
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torchlsq.quantized import LSQFakeQuantizer, enable_rank_sync
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU())
    m.qconfig = QConfig(activation=LSQFakeQuantizer.with_args(observer=MovingAverageMinMaxObserver, otype="activation", init_batches=1),
                        weight=LSQFakeQuantizer.with_args(observer=MovingAveragePerChannelMinMaxObserver, otype="weight", dtype=torch.qint8,
                                                          qscheme=torch.per_channel_symmetric))
    prepare_qat(m.train(), inplace=True)
    x = torch.randn(2, 3, 8, 8)
    for _ in range(3):
        m(x)
    assert len(enable_rank_sync(m)) == 1
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert torch.equal(m2(x), m(x))
    assert sorted(q._sync for q in m2.modules() if isinstance(q, LSQFakeQuantizer)) == [False, True]
    assert all(q._group_ref.group is None for q in m2.modules() if isinstance(q, LSQFakeQuantizer))
    m3 = copy.deepcopy(m)
    assert torch.equal(m3(x), m(x))
