#!/usr/bin/env python3
"""Is one rank's cfg4 shard step GPU-bound?  (diagnostic; 1 GPU)

BASELINE config 4 at 8 GPUs gives every rank a [128,1024,14,14] shard: ~33 us forward + ~53 us backward of GPU work per
step.  The step is only as fast as the host can enqueue it, so this measures, for the two host layers above the C ABI
(the C++ torch binding and the Python torch.library/ctypes registration):
  * back-to-back steps (forward op + sharded `*_wide` backward op + the rounding of the fp64 sums), no sync between
    steps: wall time per step;
  * the same step's host cost alone (a tiny tensor: GPU time negligible);
  * the GPU-side time of the step (20 steps captured in one HIP graph).
The collective itself (one 16-byte RCCL all-reduce per step, issued asynchronously) cannot run on a 1-GPU box.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))


def main():
    import torch
    import torchlsq  # noqa: F401
    from torchlsq import extension, synth
    dev = torch.device("cuda:0")
    c = synth.CONFIGS["cfg4"]
    q = (c["qmin"], c["qmax"], c["tmin"], c["tmax"])
    tail = q + (True, 1.0, False, False, False)
    out = {}
    for label, shape in (("cfg4_shard_8gpu", [128, 1024, 14, 14]), ("tiny_host_only", [1, 16, 14, 14])):
        x, g, scale, shift = synth.make_inputs("cfg4", device=dev, dtype=torch.float32, shape=shape)
        n_global = x.numel() * 8
        res = {"shape": shape, "n": x.numel()}
        for binding in ("native", "ctypes"):
            try:
                extension.set_host_binding(binding)
            except RuntimeError as e:
                res[binding] = {"error": str(e)}
                continue
            ops = torch.ops.torchlsq_native if binding == "native" else torch.ops.torchlsq

            def step():
                y = ops.lsq_forward_per_tensor(x, scale, shift, *tail)
                dx, wide = ops.lsq_backward_per_tensor_wide(g, x, scale, shift, *tail, n_global)
                return wide.to(torch.float32)
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            steps = 2000
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            t_enq = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            res[binding] = {"us_per_step_wall": round(t_all / steps * 1e6, 2), "us_per_step_enqueue": round(t_enq / steps * 1e6, 2)}
        # GPU-side: 20 steps in one graph
        extension.set_host_binding("native")
        ops = torch.ops.torchlsq_native
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(3):
                ops.lsq_forward_per_tensor(x, scale, shift, *tail)
                ops.lsq_backward_per_tensor_wide(g, x, scale, shift, *tail, n_global)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(20):
                    y = ops.lsq_forward_per_tensor(x, scale, shift, *tail)
                    dx, wide = ops.lsq_backward_per_tensor_wide(g, x, scale, shift, *tail, n_global)
                    w32 = wide.to(torch.float32)
            gr.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gr.replay()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        ts.sort()
        res["graph_us_per_step"] = round(ts[len(ts) // 2], 2)
        for b in ("native", "ctypes"):
            if "us_per_step_wall" in res.get(b, {}):
                res[b]["eager_over_graph_rate"] = round(res["graph_us_per_step"] / res[b]["us_per_step_wall"], 4)
        out[label] = res
        print(label, json.dumps(res))
    return out


if __name__ == "__main__":
    main()
