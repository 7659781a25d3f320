#!/usr/bin/env python3
"""bench.py -- GElem/s of the LSQ fake-quantize hot path (forward op + backward op) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one `lsq_forward_per_tensor` + one `lsq_backward_per_tensor` (training mode) over one
batch of synthetic input already resident in HBM: BASELINE.json config 2, per-tensor quint8,
fp32 [128,512,56,56] (205.5 M elements, 822 MB per tensor) PER GPU.  With N > 1 the batch is sharded
across the ranks (weak scaling: every rank owns a [128,512,56,56] shard of a [128*N,512,56,56] batch),
the backward uses the GLOBAL element count in the gradient scaler and ONE RCCL all-reduce of the
packed fp64 [d_scale, d_shift] pair per step -- inside the timed region.

Rank 0 prints ONE JSON line; `value` is the whole-job aggregate: (elements of all ranks * K) / time,
time = max over ranks of the K-step wall time bracketed by barrier + synchronize.

Extra objects on the same line:
  roofline      HBM roofline of the dominant kernel (the fused backward, 12 algorithmic bytes/element:
                read grad + read x + write dx), from its average launch duration measured live with
                HIP events on the launch stream inside the timed region.  `fwd` carries the same for
                the forward kernel (8 B/element) and `step_frac` the 20 B/element fwd+bwd figure
                BASELINE.md quotes the 70 % target on.
  cpu_baseline  the reference's own CPU csrc (oracle/_ref/libtorchlsq_ref_ops.so, kind "reference") --
                or, if that build is absent, the C restatement (kind "port") -- timed on this box's
                host cores on a bounded sample, rank 0 at N = 1 only.  A reported baseline, not the target.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_FWD, BYTES_BWD = 8, 12   # algorithmic bytes per fp32 element


def cpu_baseline(sample_shape, reps):
    """Time the reference CPU path (or the port) on a bounded sample of the same workload."""
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--shape",
           ",".join(str(s) for s in sample_shape), "--reps", str(reps)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": (out.stderr or out.stdout)[-400:]}
    except Exception as e:  # never let the reported baseline break the bench line
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", help="cfg2 (default, weak-scaled per GPU) | cfg4 (strong: "
                    "[1024,1024,14,14] split over the ranks) | cfg1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)            # "gloo" + --single-device: smoke-test
    ap.add_argument("--single-device", action="store_true", help=argparse.SUPPRESS)  # the N>1 control flow on a 1-GPU box
    ap.add_argument("--variant-fwd", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--variant-bwd", type=int, default=0, help=argparse.SUPPRESS)
    a = ap.parse_args()

    # multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); the
    # launcher normally exports it already
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import torchlsq  # noqa: F401
    from torchlsq import extension, synth
    from torchlsq.distributed import sharded_backward

    extension._assert_has_ops()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    dev = torch.device("cuda", 0 if a.single_device else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" == RCCL on ROCm
        else:
            dist.init_process_group(backend=a.backend)

    c = synth.CONFIGS[a.workload]
    shape = list(c["shape"])
    scaling = "weak"
    if a.workload == "cfg4":           # strong scaling: fixed global batch split over the ranks
        assert shape[0] % world == 0
        shape[0] //= world
        scaling = "strong"
    x, g, scale, shift = synth.make_inputs(a.workload, device=dev, dtype=torch.float32, shape=shape)
    n_local = x.numel()
    n_global = n_local * world
    ops = torch.ops.torchlsq
    q = (c["qmin"], c["qmax"], c["tmin"], c["tmax"])
    sym = not c["affine"]

    def fwd():
        if a.variant_fwd == 0:     # the registered op: dispatcher -> ctypes -> C ABI -> kernel
            return ops.lsq_forward_per_tensor(x, scale, shift, *q, True, 1.0, sym, False, False)
        return extension.hip_forward_per_tensor(x, scale, shift, *q, True, 1.0, sym, False, False, variant=a.variant_fwd)

    pending = []   # N > 1: the previous step's in-flight all-reduce (RCCL runs it on its own stream)

    def bwd():
        if world == 1:
            if a.variant_bwd == 0:
                return ops.lsq_backward_per_tensor(g, x, scale, shift, *q, True, 1.0, sym, False, False)
            return extension.hip_backward_per_tensor(g, x, scale, shift, *q, True, 1.0, sym, False, False,
                                                     variant=a.variant_bwd)
        # batch-sharded: local fused backward with the GLOBAL numel in the gradient scaler, then ONE
        # all-reduce of the packed fp64 [d_scale, d_shift] pair.  The collective is issued async and
        # consumed one step later (d_scale/d_shift are only needed by the optimizer), so its latency
        # hides behind the next step's kernels; every reduction is completed inside the timed region.
        dx, wide, work = sharded_backward(g, x, scale, shift, *q, 1, True, 1.0, c["affine"], False, False, False,
                                          None, n_global, async_op=True)
        drain()
        pending.append((wide, work))
        return dx

    def drain():
        while pending:
            wide, work = pending.pop()
            work.wait()                                   # stream-level wait, the host does not block
            ds_db = wide.to(torch.float32)                # the rounding to the parameter type
        return None

    for _ in range(a.warmup):
        y = fwd()
        r = bwd()
    drain()
    torch.cuda.synchronize()

    # HIP events bracket the forward and the backward op on every `stride`-th step of the timed region (an
    # event record drains the queue, so bracketing every op of every step would itself cost a few % of a
    # 0.7 ms step); at least 10 steps are sampled.
    stride = max(1, min(4, a.steps // 10))
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] if i % stride == 0 else None for i in range(a.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        e = ev[i]
        if e is None:
            y = fwd()
            r = bwd()
        else:
            e[0].record()
            y = fwd()
            e[1].record()
            r = bwd()
            e[2].record()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())

    fwd_ms = sorted(e[0].elapsed_time(e[1]) for e in ev if e is not None)
    bwd_ms = sorted(e[1].elapsed_time(e[2]) for e in ev if e is not None)
    fwd_avg = sum(fwd_ms) / len(fwd_ms)
    bwd_avg = sum(bwd_ms) / len(bwd_ms)

    if rank == 0:
        value = n_global * a.steps / elapsed_max / 1e9
        bwd_gbs = BYTES_BWD * n_local / (bwd_avg * 1e-3) / 1e9
        fwd_gbs = BYTES_FWD * n_local / (fwd_avg * 1e-3) / 1e9
        step_gbs = (BYTES_FWD + BYTES_BWD) * n_local / ((fwd_avg + bwd_avg) * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.isfile(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("workload") == a.workload and tj.get("n_local") == n_local:
                    traffic = tj.get("bwd_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "GElem/s fake-quant fwd+bwd, per-tensor int8, 1/2/4/8 MI355X; % HBM roofline",
            "value": round(value, 3), "unit": "GElem/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed_max / a.steps * 1e3, 5), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: per-tensor quint8 (qmin,qmax=%d,%d) fp32 %s per GPU, lsq_forward_per_tensor + "
                                   "lsq_backward_per_tensor%s" % (a.workload, c["qmin"], c["qmax"], shape,
                                                                  "" if world == 1 else ", batch-sharded, 1 RCCL all-reduce of fp64 [ds,db] per step"),
                       "elements_per_gpu": n_local, "global_elements": n_global,
                       "parallelism": "dp%d" % world},
            "roofline": {"bound": "hbm", "kernel": "lsq::bwd_pt_kernel<io_f32> (fused dx + d_scale/d_shift reduction)",
                         "achieved": round(bwd_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(bwd_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bytes_per_launch": BYTES_BWD * n_local, "avg_launch_ms": round(bwd_avg, 5), "launches_timed": len(bwd_ms),
                         "median_launch_ms": round(bwd_ms[len(bwd_ms) // 2], 5),
                         "fwd": {"kernel": "lsq::fwd_pt_kernel<io_f32>", "achieved": round(fwd_gbs, 1),
                                 "frac": round(fwd_gbs / HBM_PEAK_GBS, 4), "avg_launch_ms": round(fwd_avg, 5),
                                 "bytes_per_launch": BYTES_FWD * n_local},
                         "step_achieved": round(step_gbs, 1), "step_frac": round(step_gbs / HBM_PEAK_GBS, 4),
                         # the read-only variant next to the all-traffic figure -- 12 of the 20
                         # algorithmic bytes per element are reads (4 forward + 8 backward), so it is 0.6 x step_frac
                         "step_reads_only_achieved": round(step_gbs * 12.0 / 20.0, 1),
                         "step_reads_only_frac": round(step_gbs * 12.0 / 20.0 / HBM_PEAK_GBS, 4)},
        }
        if world == 1:
            # Context for the roofline fraction, measured live on THIS box after the timed region: the framework's /
            # vendor's own kernels on the same two traffic shapes (ATen's vectorised add = 2 reads : 1 write like the
            # backward; its copy = 1 read : 1 write like the forward).  Not part of `value`.
            try:
                def _gbs(fn, bytes_per_elem, reps=5):
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        fn()
                    e1.record()
                    e1.synchronize()
                    return round(bytes_per_elem * n_local / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9, 1)
                scratch = torch.empty_like(x)
                line["roofline"]["same_box_reference_kernels"] = {
                    "aten_add_2r1w_GBps": _gbs(lambda: torch.add(g, x, out=scratch), BYTES_BWD),
                    "aten_copy_1r1w_GBps": _gbs(lambda: scratch.copy_(x), BYTES_FWD)}
                del scratch
            except Exception as e:      # context only: never let it break the bench line
                line["roofline"]["same_box_reference_kernels"] = {"error": repr(e)}
        if world == 1 and not a.no_cpu_baseline:
            sample = [max(1, shape[0] // 8)] + shape[1:]
            line["cpu_baseline"] = cpu_baseline(sample, reps=5)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
