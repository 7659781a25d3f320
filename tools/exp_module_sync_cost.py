#!/usr/bin/env python3
"""What rank sync costs a quantizer call on one GPU: `LSQFakeQuantizer(sync=True)` against the plain module, forward +
backward per step, eager, wall clock -- with the collectives executed by RCCL in a world of ONE (every all-reduce an identity,
so the numbers contain the call path, the extra launches and RCCL's enqueue + kernel, not a transport between GPUs) and the
module told it has peers.  Observer-driven init step (statistics -> packed MIN all-reduce -> one-launch tail -> eval-mode
fake-quant) and LSQ step (forward, sharded backward with the count in the collective).  Output: profiles/r0x_module_sync_cost.txt (round 5: the collectives go through the library's own RCCL communicator;
TORCHLSQ_COLLECTIVE=c10d for torch.distributed)"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29671")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torchlsq  # noqa: E402,F401
from torch.ao.quantization.observer import MovingAverageMinMaxObserver, MovingAveragePerChannelMinMaxObserver  # noqa: E402
from torchlsq import synth, distributed as D  # noqa: E402
from torchlsq.quantized import LSQFakeQuantizer  # noqa: E402
from torchlsq.quantized.modules import observers as OBS  # noqa: E402


def step_time(m, xs, ws, steps=200):
    def one(k):
        x = xs[k % len(xs)]
        x.grad = None
        y = m(x)
        if y.requires_grad:
            m.scale.grad = None
            m.shift.grad = None
            (y * ws[k % len(ws)]).sum().backward()
    for k in range(30):
        one(k)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for k in range(steps):
            one(k)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    return best


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    print("# tools/exp_module_sync_cost.py on one MI355X: us per module call (forward + backward incl. the loss's mul / sum / their backward), eager, best of 3 x 200 steps")
    print("# sync = LSQFakeQuantizer(sync=True) with RCCL (world of one, identity all-reduces) told it has a peer; plain = the same module without")
    for shape in ((128, 1024, 14, 14), (32, 256, 28, 28), (4, 64, 56, 56)):
        n = 1
        for d in shape:
            n *= d
        K = max(2, min(8, (1 << 30) // (n * 8)))
        xs = [synth.normal_like(n, 10 + k, 0.8, 1.0, device=dev).view(shape).requires_grad_(True) for k in range(K)]
        ws = [synth.normal_like(n, 40 + k, 0.0, 1.0, device=dev).view(shape) for k in range(K)]
        for obs_cls, extra, name in ((MovingAverageMinMaxObserver, {}, "per-tensor"),
                                     (MovingAveragePerChannelMinMaxObserver, dict(qscheme=torch.per_channel_affine, ch_axis=1), "per-channel")):
            row = []
            for phase in ("observer init step", "LSQ step"):
                times = {}
                for sync in (False, True):
                    OBS._dist_world = (lambda group: 2) if sync else (lambda group: 1)
                    D.assume_peers(sync)          # ... and torchlsq.distributed issues its collectives as if the group had peers
                    m = LSQFakeQuantizer(obs_cls, "activation", init_batches=(10 ** 9 if phase.startswith("observer") else 0), sync=sync, **extra).train()
                    m(xs[0])
                    m.to(dev)
                    times[sync] = step_time(m, xs, ws)
                row.append("%s: plain %6.1f  sync %6.1f  (+%.1f)" % (phase, times[False], times[True], times[True] - times[False]))
            print("%-20s %-12s %s" % ("x".join(str(d) for d in shape), name, "   |   ".join(row)), flush=True)
        del xs, ws
        torch.cuda.empty_cache()
    D.assume_peers(False)
    D.destroy_native_comms()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
