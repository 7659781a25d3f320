"""Worker processes of the rank-synchronised `LSQFakeQuantizer` tests (tests/test_module_sync_cpu.py on CPU over gloo,
tests/test_module_sync_gpu.py: the same ranks sharing the one GPU of the test box, collectives over gloo).

`replay` drives the scenarios of tests/golden/make_module_traces.py -- the ones the REFERENCE module was traced on, on one
device and the whole batch -- with the batch (dim 0) split over the ranks, and compares every rank after every call with
  * the reference trace: scale / shift lists EXACTLY, state flags, requires_grad flags, gradients like test_host_logic.py;
  * this repository's module run UNSHARDED on the whole batch in the same process: y / dx of the shard bit-exact,
    scale.grad / shift.grad within 1e-6 of sum|terms| (helpers.py bar), scale / shift bit-identical;
  * the collective count: one all-reduce per observer step, one per LSQ backward, nothing else.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

SYNC_SCENARIOS = ("act_observer_pt", "act_learnable_pt", "act_observer_pc", "act_fakequant_only", "act_8bit_custom_range",
                  "act_symmetric_pt", "act_eval_midway", "act_toggle_learning", "act_disable_fake_quant", "weight_pc_sym")


def _setup_paths():
    for p in (os.path.join(ROOT, "lsqfakequantize-pytorch_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _driver():
    spec = importlib.util.spec_from_file_location("_trace_driver", os.path.join(GOLDEN, "make_module_traces.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def shard_bounds(rows, world, uneven):
    """row boundaries of the shards; uneven: different sizes and (world 4) one EMPTY shard"""
    if not uneven:
        step = -(-rows // world)
        return [min(r * step, rows) for r in range(world + 1)]
    if world == 4:
        return [0, (rows + 1) // 2, (rows + 1) // 2, rows - 1, rows]
    cuts = sorted(set([0, rows] + [max(1, (rows * (r * r + 1)) // (world * world + 1)) for r in range(1, world)]))
    while len(cuts) < world + 1:
        cuts.append(rows)
    return cuts


def replay(rank, world, port, uneven, device, out_q, names=SYNC_SCENARIOS, sync=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    _setup_paths()
    import torchlsq  # noqa: F401
    from torch.ao.quantization import observer as obs_mod
    from torchlsq.quantized import LSQFakeQuantizer
    drv = _driver()
    S = drv.S
    dev = torch.device(device)
    with open(os.path.join(GOLDEN, "module_traces.json")) as f:
        traces = json.load(f)["traces"]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    problems = []
    try:
        calls = {"n": 0}
        real_all_reduce = dist.all_reduce

        def counting_all_reduce(*a, **k):
            calls["n"] += 1
            return real_all_reduce(*a, **k)
        dist.all_reduce = counting_all_reduce

        for name in names:
            t = traces[name]
            sc = t["scenario"]
            observer = getattr(obs_mod, sc["observer"]) if sc["observer"] else None
            kw = drv.build_kwargs(sc["ctor"])
            m = LSQFakeQuantizer(observer, sync=sync, **kw).train()          # this rank's replica
            full = LSQFakeQuantizer(observer, **kw).train()                  # the unsharded module on the whole batch
            is_weight = sc["ctor"]["otype"] == "weight"
            shape = sc["shape"]
            n = int(np.prod(shape))
            b = shard_bounds(shape[0], world, uneven)
            sl = slice(0, shape[0]) if is_weight else slice(b[rank], b[rank + 1])     # weights are replicated, not sharded
            for i in range(sc["calls"]):
                want = t["calls"][i]
                tag = "%s call %d rank %d" % (name, i, rank)
                act = sc.get("actions", {}).get(str(i))
                if act:
                    getattr(m, act)()
                    getattr(full, act)()
                if sc.get("eval_from") is not None and i >= sc["eval_from"]:
                    m.eval()
                    full.eval()
                x = S.normal_like(n, 100 + i, sc["x_mean"], sc["x_std"]).view(shape).to(dev)
                w = S.normal_like(n, 200 + i, 0.0, 1.0).view(shape).to(dev)
                xs = x[sl].clone().requires_grad_(True)
                xf = x.clone().requires_grad_(True)
                before = calls["n"]
                ys = m(xs)
                yf = full(xf)
                if i == 0 and dev.type == "cuda":      # parameters exist now: the module moves to the GPU like a user's would
                    m.to(dev)
                    full.to(dev)
                backward_ran = bool(ys.requires_grad)
                if backward_ran:
                    for mod in (m, full):
                        for prm in (mod.scale, mod.shift):
                            if prm is not None:
                                prm.grad = None
                    (ys * w[sl]).sum().backward()
                    (yf * w).sum().backward()
                used = calls["n"] - before

                def lst(v):
                    return None if v is None else [float(a) for a in v.detach().reshape(-1).tolist()]
                # 1. against the reference module's trace (whole batch, one device)
                # (CPU: exactly.  GPU: to an ulp -- the trace is the reference on the CPU, and torch's GPU kernels divide by a
                #  scalar as a multiplication with its reciprocal, which the observer's qparams -- torch's own arithmetic on
                #  whichever device -- inherit; the unsharded module on the same device, below, must match exactly)
                if dev.type == "cpu":
                    same = lst(m.scale) == want["scale"] and lst(m.shift) == want["shift"]
                else:
                    same = (np.allclose(lst(m.scale), want["scale"], rtol=3e-7, atol=0) and
                            np.allclose(lst(m.shift), want["shift"], rtol=3e-7, atol=1e-9))
                if not same:
                    problems.append(tag + ": scale/shift differ from the reference trace: %r vs %r" % (lst(m.scale), want["scale"]))
                for k, got in (("current_batch", int(m.current_batch[0])), ("observer_enabled", int(m.observer_enabled[0])),
                               ("fake_quant_enabled", int(m.fake_quant_enabled[0])), ("learning_enabled", int(m.learning_enabled[0])),
                               ("scale_requires_grad", bool(m.scale.requires_grad)), ("shift_requires_grad", bool(m.shift.requires_grad)),
                               ("y_is_x", bool(ys is xs))):
                    if got != want[k]:
                        problems.append(tag + ": %s %r, reference trace %r" % (k, got, want[k]))
                for pname in ("scale", "shift"):
                    got, ref = getattr(m, pname).grad, want[pname + "_grad"]
                    if (got is None) != (ref is None):
                        problems.append(tag + ": %s.grad presence differs from the reference trace" % pname)
                    # (a WIRING check against the reference module's trace: its gradients come out of the reference's fp32
                    #  at::sum over mixed-sign terms, whose own cancellation error is ~1e-5 of the sum; the arithmetic bar -- 1e-6 of
                    #  sum|terms| against the fp64 oracle -- is enforced in tests/test_oracle_pinned.py and tests/test_sharded_*.py)
                    elif got is not None and not np.allclose(np.array(lst(got)), np.array(ref), rtol=1e-4, atol=1e-8):
                        problems.append(tag + ": %s.grad %r vs reference trace %r" % (pname, lst(got), ref))
                # 2. against this repository's module on the whole batch
                if not torch.equal(ys.detach(), yf.detach()[sl]):
                    problems.append(tag + ": y of the shard differs from the unsharded y")
                if backward_ran and xs.grad is not None and not torch.equal(xs.grad, xf.grad[sl]):
                    problems.append(tag + ": dx of the shard differs from the unsharded dx")
                if not (torch.equal(m.scale, full.scale) and torch.equal(m.shift, full.shift)):
                    problems.append(tag + ": scale/shift differ from the unsharded module's")
                for pname in ("scale", "shift"):
                    g1, g2 = getattr(m, pname).grad, getattr(full, pname).grad
                    if g1 is not None and g2 is not None:
                        # bar: 1e-6 of sum|terms|; the terms are bounded by |w| * (quant range) / sqrt(numel * qmax) per element
                        tol = 1e-6 * float(w.abs().sum()) * 256.0 + 1e-12
                        if float((g1.double() - g2.double()).abs().max()) > tol:
                            problems.append(tag + ": %s.grad %r vs unsharded %r" % (pname, lst(g1), lst(g2)))
                        rel = float(((g1.double() - g2.double()).abs() / g2.double().abs().clamp_min(1e-30)).max())
                        if rel > 2e-5:      # small mixed-sign sums: cancellation amplifies the 1e-7 rounding of the scaler
                            problems.append(tag + ": %s.grad relative error %g vs the unsharded module" % (pname, rel))
                # 3. collectives: one per observer step (the observer ran iff it is still enabled after the call), one per
                #    LSQ backward (full LSQ <=> scale.requires_grad), none for weights / the creating call / plain fake-quant
                expect = 0
                if not is_weight and i > 0:
                    expect += int(want["observer_enabled"] == 1)
                    expect += int(backward_ran and want["scale_requires_grad"])
                if used != expect:
                    problems.append(tag + ": %d collectives, expected %d" % (used, expect))
        out_q.put((rank, problems))
    except Exception as e:     # noqa: BLE001  (report instead of hanging the other ranks' queue reader)
        import traceback
        out_q.put((rank, problems + ["rank %d raised %r\n%s" % (rank, e, traceback.format_exc())]))
    finally:
        dist.destroy_process_group()


def ddp_train(rank, world, port, device, grads, out_q, steps=6):
    """A small QAT model under DistributedDataParallel (prepare_ddp) next to the same model trained in ONE process on the
    whole batch: the replicas must stay bit-identical to each other and follow the single-process trajectory."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    _setup_paths()
    import torchlsq  # noqa: F401
    from torch.ao.quantization import QConfig
    from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver
    from torch.nn.parallel import DistributedDataParallel as DDP
    from torchlsq.quantized import LSQFakeQuantizer, prepare_ddp
    from torchlsq import synth as S
    dev = torch.device(device)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    problems = []
    try:
        def build():
            torch.manual_seed(1234)
            model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 3, padding=1))
            act = LSQFakeQuantizer.with_args(observer=MovingAverageMinMaxObserver, otype="activation", init_batches=2)
            wgt = LSQFakeQuantizer.with_args(observer=MovingAveragePerChannelMinMaxObserver, otype="weight", dtype=torch.qint8,
                                             qscheme=torch.per_channel_symmetric)
            model.qconfig = QConfig(activation=act, weight=wgt)
            torch.ao.quantization.prepare_qat(model.train(), inplace=True)
            return model.to(dev)

        B = 4 * world

        def batch(i):
            n = B * 3 * 8 * 8
            return (S.normal_like(n, 500 + i, 0.2, 1.0).view(B, 3, 8, 8).to(dev), S.normal_like(B * 4 * 8 * 8, 600 + i, 0.0, 1.0).view(B, 4, 8, 8).to(dev))

        single, repl = build(), build()
        x0, _ = batch(0)
        single(x0)                                   # creating call: the quantizers' parameters exist afterwards
        repl(x0[rank * 4:(rank + 1) * 4])
        quantizers = [m for m in repl.modules() if isinstance(m, LSQFakeQuantizer)]
        refresh = {"n": 0}
        for q in quantizers:
            orig = q._refresh_host_state

            def counted(orig=orig):
                refresh["n"] += 1
                return orig()
            q._refresh_host_state = counted
        if rank != 0:      # a replica that did NOT start identical (e.g. a checkpoint restored on rank 0 only): prepare_ddp
            with torch.no_grad():     # broadcasts what it tells DDP to leave alone
                for q in quantizers:
                    if q.dtype == torch.quint8:
                        q.scale.mul_(1.7)
                        q.shift.add_(0.3)
                        q.current_batch[0] = 1        # ... including the state flags: left alone, this rank would leave the
                                                      # init phase one batch early and the ranks' collectives would not pair up
        calls = {"n": 0}
        real_all_reduce = dist.all_reduce

        def counting_all_reduce(*a, **k):
            calls["n"] += 1
            return real_all_reduce(*a, **k)
        dist.all_reduce = counting_all_reduce
        # 'mean': the module reduces its gradients itself (its parameters are on DDP's ignore list); 'ddp': DDP does -- and has to
        # be told that parameters may go without a gradient (the observer-driven init batches run the quantizers in eval mode)
        ddp = DDP(prepare_ddp(repl, grads=grads), find_unused_parameters=(grads == "ddp"))
        opt_s = torch.optim.SGD(single.parameters(), lr=0.05)
        opt_r = torch.optim.SGD(ddp.parameters(), lr=0.05)
        refresh["n"] = 0
        for i in range(1, steps + 1):
            x, tgt = batch(i)
            opt_s.zero_grad(set_to_none=True)
            ((single(x) - tgt) ** 2).mean().backward()           # loss = mean over the WHOLE batch
            opt_s.step()
            sl = slice(rank * 4, (rank + 1) * 4)
            opt_r.zero_grad(set_to_none=True)
            ((ddp(x[sl]) - tgt[sl]) ** 2).mean().backward()      # each rank: mean over its shard; DDP averages the ranks
            opt_r.step()
            flat = torch.cat([p.detach().reshape(-1).double() for p in repl.parameters()])
            gathered = [torch.empty_like(flat) for _ in range(world)]
            dist.all_gather(gathered, flat)
            if i == 1:
                refresh["n"] = 0        # (the first call after .to(device) re-reads the moved buffers once: not DDP's doing)
            if i == 3:
                calls["n"] = 0          # the observer-driven init batches (2) are over: LSQ steps from here on
            if any(not torch.equal(g, gathered[0]) for g in gathered):
                problems.append("step %d: replicas differ across ranks" % i)
            for (n1, p1), (_, p2) in zip(single.named_parameters(), repl.named_parameters()):
                if not torch.allclose(p1, p2, rtol=2e-4, atol=1e-6):
                    problems.append("step %d: %s diverges from the single-process run: max |diff| %g" %
                                    (i, n1, float((p1 - p2).abs().max())))
        # explicit all-reduces of the module in the LSQ steps: one per activation quantizer and backward ('mean'), none ('ddp')
        n_act = sum(1 for q in quantizers if q.dtype == torch.quint8)
        want = 0 if grads == "ddp" else n_act * (steps - 3)
        if calls["n"] != want:
            problems.append("%d explicit all-reduces in the LSQ steps, expected %d (grads=%r)" % (calls["n"], want, grads))
        if refresh["n"] != 0:
            problems.append("the state flags were re-read from the device %d times under DDP (buffer broadcasts not ignored)" % refresh["n"])
        out_q.put((rank, problems))
    except Exception as e:     # noqa: BLE001
        import traceback
        out_q.put((rank, problems + ["rank %d raised %r\n%s" % (rank, e, traceback.format_exc())]))
    finally:
        dist.destroy_process_group()


def nan_sync(rank, world, port, device, out_q):
    """A NaN in ONE rank's shard of an observer-driven init batch (the last row of the batch: the last rank's): every rank must
    end up with the parameters the REFERENCE module has after seeing the whole batch (tests/golden/module_traces.json,
    extras.nan_in_the_batch: torch.aminmax turns both extremes NaN) -- the same on all ranks, bit for bit.  What a NaN does in
    a MIN all-reduce is the backend's business, so the collective never sees one (torchlsq.distributed.all_reduce_minmax)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    _setup_paths()
    import torchlsq  # noqa: F401
    from torchlsq.quantized import LSQFakeQuantizer
    drv = _driver()
    with open(os.path.join(GOLDEN, "module_traces.json")) as f:
        want_all = json.load(f)["extras"]["nan_in_the_batch"]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    problems = []
    try:
        b = shard_bounds(4, world, False)
        for kind, per_channel in (("per_tensor", False), ("per_channel", True)):
            got = drv.nan_in_the_batch(LSQFakeQuantizer, per_channel, device=device, rows=(b[rank], b[rank + 1]),
                                       sync_kwargs=dict(sync=True))
            for g, w in zip(got, want_all[kind]):
                tag = "%s call %d rank %d" % (kind, g["call"], rank)
                for k in ("scale", "shift"):
                    gv = np.array([float(v) for v in g[k]])
                    wv = np.array([float(v) for v in w[k]])
                    if not np.array_equal(np.isnan(gv), np.isnan(wv)):
                        problems.append(tag + ": NaN pattern of %s %r, reference %r" % (k, g[k], w[k]))
                    elif device == "cpu":
                        if not np.array_equal(gv[~np.isnan(gv)], wv[~np.isnan(wv)]):
                            problems.append(tag + ": %s %r, reference %r" % (k, g[k], w[k]))
                    elif not np.allclose(gv[~np.isnan(gv)], wv[~np.isnan(wv)], rtol=3e-7, atol=1e-9):
                        problems.append(tag + ": %s %r, reference %r" % (k, g[k], w[k]))
                for k in ("observer_enabled", "current_batch"):
                    if g[k] != w[k]:
                        problems.append(tag + ": %s %r, reference %r" % (k, g[k], w[k]))
            # replicas: the same bits everywhere (NaNs included: as text)
            mine = json.dumps([(g["scale"], g["shift"]) for g in got])
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
            if any(e != everyone[0] for e in everyone):
                problems.append("%s: the replicas differ across ranks" % kind)
        out_q.put((rank, problems))
    except Exception as e:     # noqa: BLE001
        import traceback
        out_q.put((rank, problems + ["rank %d raised %r\n%s" % (rank, e, traceback.format_exc())]))
    finally:
        dist.destroy_process_group()
