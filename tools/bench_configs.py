#!/usr/bin/env python3
"""Time forward / backward ops of every BASELINE.json config on the MI355X (diagnostic tool).

For each config: median HIP-event time of the forward op and of the backward op through
torch.ops.torchlsq (product path), GElem/s fwd+bwd and the algorithmic-bytes fraction of 8 TB/s.
Small configs are measured "hot" (same buffers every launch: served by the 256 MB Infinity Cache) and
"cold-rotated" (a ring of buffer sets larger than the cache) -- SURVEY.md section 8(d).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lsqfakequantize-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--configs", default="cfg1,cfg2,cfg3,cfg4s,cfg5,cfg5_bf16,cfg5_axis0")
    ap.add_argument("--pc-bpc", type=int, default=0, help="override workgroups-per-CU of the per-channel kernels")
    ap.add_argument("--graph-only", action="store_true")
    a = ap.parse_args()
    import torch
    import torchlsq  # noqa: F401
    from torchlsq import synth
    ops = torch.ops.torchlsq
    from torchlsq import extension as E
    import lsq_tools  # noqa: E402  (tools build of the library: `_ex` entry points, lsq_hip_debug_* knobs)
    lsq_tools.activate()
    pcv = (4 | (1 << 8) | (1 << 9) | (a.pc_bpc << 16)) if a.pc_bpc else 0
    dev = torch.device("cuda:0")
    out = {}
    for name in a.configs.split(","):
        key = {"cfg4s": "cfg4", "cfg5_bf16": "cfg5", "cfg5_axis0": "cfg5"}.get(name, name)
        c = dict(synth.CONFIGS[key])
        shape = list(c["shape"])
        if name == "cfg4s":
            shape[0] //= 8            # one rank's shard at 8 GPUs
        if name == "cfg5_axis0":
            c["axis"] = 0
        dt = torch.bfloat16 if name == "cfg5_bf16" else torch.float32
        esz = 2 if dt == torch.bfloat16 else 4
        n = 1
        for d in shape:
            n *= d
        ring = max(2, min(16, int(1.2e9 // (3 * n * esz)) + 1)) if 3 * n * esz < 1e9 else 1
        sets = []
        for r in range(ring):
            x, g, scale, shift = synth.make_inputs(c, device=dev, dtype=dt, shape=shape)
            sets.append((x, g, scale, shift))
        q = (c["qmin"], c["qmax"], c["tmin"], c["tmax"])
        sym = not c["affine"]

        def fwd(s):
            x, g, scale, shift = s
            if c["per_channel"]:
                if pcv:
                    return E.hip_forward_per_channel(x, scale, shift, c["axis"], *q, True, 1.0, sym, False, False, variant=pcv)
                return ops.lsq_forward_per_channel(x, scale, shift, c["axis"], *q, True, 1.0, sym, False, False)
            return ops.lsq_forward_per_tensor(x, scale, shift, *q, True, 1.0, sym, False, False)

        def bwd(s):
            x, g, scale, shift = s
            if c["per_channel"]:
                if pcv:
                    return E.hip_backward_per_channel(g, x, scale, shift, c["axis"], *q, True, 1.0, sym, False, False, variant=pcv)
                return ops.lsq_backward_per_channel(g, x, scale, shift, c["axis"], *q, True, 1.0, sym, False, False)
            return ops.lsq_backward_per_tensor(g, x, scale, shift, *q, True, 1.0, sym, False, False)

        res = {"shape": shape, "n": n, "dtype": str(dt), "ring": ring}
        for mode in (() if a.graph_only else (("hot", "cold") if ring > 1 else ("hot",))):
            for kind, fn, nb in (("fwd", fwd, 2 * esz * n), ("bwd", bwd, 3 * esz * n)):
                for i in range(5):
                    fn(sets[i % ring])
                torch.cuda.synchronize()
                ts = []
                for i in range(a.iters):
                    s = sets[(i % ring) if mode == "cold" else 0]
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn(s)
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1))
                ts.sort()
                med = ts[len(ts) // 2]
                res["%s_%s_us" % (kind, mode)] = round(med * 1e3, 2)
                res["%s_%s_gbs" % (kind, mode)] = round(nb / med / 1e6, 1)
            tot = (res["fwd_%s_us" % mode] + res["bwd_%s_us" % mode]) * 1e-6
            res["gelems_%s" % mode] = round(n / tot / 1e9, 2)
            res["hbm_frac_%s" % mode] = round(5 * esz * n / tot / 8e12, 4)
        # GPU-side time without host launch overhead: 20 ops captured in one HIP graph, replayed
        for kind, fn, nb in (("fwd", fwd, 2 * esz * n), ("bwd", bwd, 3 * esz * n)):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                fn(sets[0])
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    for i in range(20):
                        fn(sets[i % ring])
                gr.replay()
                torch.cuda.synchronize()
                ts = []
                for _ in range(10):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    gr.replay()
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) / 20)
            ts.sort()
            res["%s_graph_us" % kind] = round(ts[len(ts) // 2] * 1e3, 2)
            res["%s_graph_gbs" % kind] = round(nb / ts[len(ts) // 2] / 1e6, 1)
        tot = (res["fwd_graph_us"] + res["bwd_graph_us"]) * 1e-6
        res["gelems_graph"] = round(n / tot / 1e9, 2)
        res["hbm_frac_graph"] = round(5 * esz * n / tot / 8e12, 4)
        out[name] = res
        print(name, json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_configs.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
