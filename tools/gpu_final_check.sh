#!/bin/bash
# the round's last check on a fresh box: smoke, the whole GPU suite, the default bench line (what the driver runs)
export TMPDIR=/tmp
O=gpurun_out/final6
mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -5 $O/smoke.txt | cut -c1-160
timeout 3000 python3 -m pytest tests -m gpu -q > $O/tests_all.txt 2>&1
tail -3 $O/tests_all.txt | cut -c1-300
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'PY'
import json
lines=[l for l in open("gpurun_out/final6/bench_default.json").read().splitlines()]
print("stdout lines:", len(lines), "last is json:", lines[-1].startswith("{"))
d=json.loads([l for l in lines if l.startswith("{")][-1])
r=d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "step_frac", r["step_frac"], "sustained", r["sustained"]["value"], r["sustained"]["step_frac"], r["sustained"]["wall_s"], "secondary_wall_s", d.get("secondary_wall_s"))
for s in d.get("secondary", []):
    print(s["workload"], s.get("value"), s.get("ms_per_step"), s.get("fwd_ms"), s.get("bwd_ms"), s.get("error", ""), s.get("wall_over_solo_shard_step", ""))
PY
