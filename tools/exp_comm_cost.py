#!/usr/bin/env python3
"""What the one collective of a sharded backward costs on ONE GPU (RCCL world of one: every all-reduce an identity) -- host time
per call and the latency it adds to a stream of small kernels -- for the library's own communicator (lsq_hip_comm_*: begin/end
on a side stream at normal / high priority, all_reduce on the caller's stream) and for torch.distributed.all_reduce.
    python tools/exp_comm_cost.py            # table -> profiles/r05_comm_cost.txt"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))


def child(prio):
    os.environ["LSQ_COMM_SIDE_PRIORITY"] = prio
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29713")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import torchlsq  # noqa: F401
    from torchlsq import distributed as D
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    D.assume_peers(True)
    comm = D.native_comm(None, dev)
    assert comm is not None
    buf = torch.ones(3, dtype=torch.float64, device=dev)
    out = torch.zeros(3, dtype=torch.float64, device=dev)
    small = torch.ones(4096, device=dev)

    def loop(fn, n=2000):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6

    pend = []

    def native_async(o=None):
        small.add_(1.0)
        if pend:
            comm.end(pend.pop())
        pend.append(comm.begin(buf, out=o))

    def native_same(o=None):
        small.add_(1.0)
        comm.all_reduce(buf, out=o)

    works = []

    def c10d_async():
        small.add_(1.0)
        if works:
            works.pop().wait()
        works.append(dist.all_reduce(buf, async_op=True))

    def c10d_sync():
        small.add_(1.0)
        dist.all_reduce(buf)

    rows = [("one small kernel alone", lambda: small.add_(1.0)),
            ("+ native begin/end, in place", native_async), ("+ native begin/end, out of place", lambda: native_async(out)),
            ("+ native all_reduce on the stream, in place", native_same), ("+ native all_reduce on the stream, out of place", lambda: native_same(out)),
            ("+ c10d all_reduce async_op, waited a step later", c10d_async), ("+ c10d all_reduce (blocking order)", c10d_sync)]
    print("## side stream priority: %s" % ("highest" if prio == "1" else "default"))
    for name, fn in rows:
        issue, wall = loop(fn)
        print("%-52s host issue %6.1f us / step    wall %6.1f us / step" % (name, issue, wall), flush=True)
    while pend:
        comm.end(pend.pop())
    while works:
        works.pop().wait()
    torch.cuda.synchronize()
    D.destroy_native_comms()
    dist.destroy_process_group()


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "breakdown"):
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        print("# tools/exp_comm_cost.py on one MI355X, RCCL world of one: a 4096-element add_ per step + the collective of a 24-byte buffer; us per step")
        print("# host issue = the Python loop's time per step (nothing waited for); wall = including the final synchronize")
        sys.stdout.flush()
        for prio in ("0", "1"):
            env = dict(os.environ, MASTER_PORT=str(29713 + int(prio)))
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", prio], env=env, check=False)


def breakdown():
    """the N > 1 bench step on a tiny tensor, per call: where the host time goes (perf_counter around every call)"""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29733")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import torchlsq  # noqa: F401
    from torchlsq import distributed as D, synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    D.assume_peers(True)
    ops = torch.ops.torchlsq_native
    for shape in ((1, 16, 14, 14), (128, 1024, 14, 14)):
        x, g, scale, shift = synth.make_inputs("cfg4", device=dev, shape=shape)
        n = x.numel()
        tail = (0, 127, 0, 255, True, 1.0, False, False, False)
        for route in ("native", "native-side-consumer", "native-inline", "c10d", "none"):
            D.set_native_collective(route.startswith("native"))
            acc = {}

            def timed(name, fn):
                t0 = time.perf_counter()
                r = fn()
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
                return r
            pending = []
            steps = 400
            for it in range(steps + 50):
                if it == 50:
                    torch.cuda.synchronize()
                    acc.clear()
                    t_all = time.perf_counter()
                timed("forward op", lambda: ops.lsq_forward_per_tensor(x, scale, shift, *tail))
                dx, wide = timed("backward_wide op", lambda: ops.lsq_backward_per_tensor_wide(g, x, scale, shift, *tail, 8 * n))
                if route == "native-side-consumer":
                    # the reduction AND its consumer (the cast) on the communicator's stream; the compute stream never waits
                    comm = D.native_comm(None, dev)
                    tk = timed("collective issue", lambda: comm.begin(wide))
                    side = comm.side_stream()
                    def consume():
                        with torch.cuda.stream(side):
                            r_ = wide.to(torch.float32)
                        wide.record_stream(side)
                        return r_
                    timed("cast (side stream)", consume)
                    last_ticket = tk
                    continue
                if route == "none":
                    work = None
                else:
                    work = timed("collective issue", lambda: D._all_reduce_sum(wide, None, async_op=(route != "native-inline")))
                if pending:
                    w_, wk = pending.pop()
                    if wk is not None:
                        timed("wait", wk.wait)
                    timed("cast", lambda: w_.to(torch.float32))
                pending.append((wide, work))
            if route == "native-side-consumer":
                comm.end(last_ticket)        # the one join
                pending = [(None, None)]
            t_issue = time.perf_counter() - t_all
            torch.cuda.synchronize()
            t_wall = time.perf_counter() - t_all
            print("%-18s %-6s issue %6.1f us/step, wall %6.1f us/step:  %s" % ("x".join(map(str, shape)), route, t_issue / steps * 1e6, t_wall / steps * 1e6,
                  "  ".join("%s %.1f" % (k, v / steps * 1e6) for k, v in acc.items())), flush=True)
            if pending and pending[0][1] is not None:
                pending[0][1].wait()
    torch.cuda.synchronize()
    D.destroy_native_comms()
    dist.destroy_process_group()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "breakdown":
    breakdown()
