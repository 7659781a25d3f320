#!/usr/bin/env python3
"""After tools/gpu_profile_round<N>.sh ran on the GPU box: copy its summaries (gpurun_out/summ<N>) into profiles/ and rebuild
profiles/traffic_latest.json from the live PMC measurements the bench lines carry.   usage: collect_round_profiles.py [round=3]"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = int(sys.argv[1]) if len(sys.argv) > 1 else 3
TAG = "r%02d_" % RND
SRC = os.path.join(ROOT, "gpurun_out", "summ%d" % RND)
DST = os.path.join(ROOT, "profiles")
for f in sorted(glob.glob(os.path.join(SRC, TAG + "*"))):
    if f.endswith(".err"):
        continue
    shutil.copy(f, os.path.join(DST, os.path.basename(f)))
tpath = os.path.join(DST, "traffic_latest.json")
doc = json.load(open(tpath))
for f in sorted(glob.glob(os.path.join(SRC, TAG + "bench_*_n1.json"))):
    lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if not lines:
        continue
    d = json.loads(lines[-1])
    r = d["roofline"]
    src = r.get("traffic_source")
    if r.get("traffic") is None or not isinstance(src, dict):
        continue
    w = os.path.basename(f)[len(TAG + "bench_"):-len("_n1.json")]
    alg = r.get("algorithmic_bytes_per_launch") or r.get("bytes_per_launch")
    e = {"n_local": d["config"].get("n_local", d["config"].get("elements_per_gpu")),
         "bwd_hbm_bytes_per_launch": r["traffic"],
         "algorithmic_bytes_per_launch": alg,
         "ratio_traffic_over_algorithmic": round(r["traffic"] / alg, 5) if alg else None,
         "kernel": r.get("kernel"),
         "source": "profiles/%s (measured live by that run)" % os.path.basename(f),
         "raw": src}
    old = doc["workloads"].get(w, {})
    for k, v in list(e.items()):
        if v is None and k in old:
            e[k] = old[k]
    if e["algorithmic_bytes_per_launch"] and e["ratio_traffic_over_algorithmic"] is None:
        e["ratio_traffic_over_algorithmic"] = round(e["bwd_hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"], 5)
    doc["workloads"][w] = e
    print(w, e["bwd_hbm_bytes_per_launch"], e["algorithmic_bytes_per_launch"], e["ratio_traffic_over_algorithmic"])
json.dump(doc, open(tpath, "w"), indent=1)
open(tpath, "a").write("\n")
