#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5c
mkdir -p $O
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'PY'
import json
lines=[l for l in open("gpurun_out/r5c/bench_default.json").read().splitlines()]
print("stdout lines:", len(lines), "last is json:", lines[-1].startswith("{"))
d=json.loads([l for l in lines if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "step_frac", d["roofline"]["step_frac"])
for s in d.get("secondary", []):
    if "cfg4" in s["workload"]:
        print(json.dumps({k: v for k, v in s.items() if k not in ("what", "value_is", "per_op_ms_from","host_us_per_step_is")}))
PY
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1
tail -8 $O/tests_all.txt | cut -c1-300
