#!/bin/bash
# GPU call S: full GPU suite with the "auto" ticket policy; default bench line
mkdir -p gpurun_out/r03s
python -m pytest tests -m gpu -q -x > gpurun_out/r03s/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r03s/pytest.log
python bench.py --no-cpu-baseline --no-measure-traffic > gpurun_out/r03s/bench_default.json 2> gpurun_out/r03s/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03s/bench_default.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["step_frac"])
for s in d["secondary"]:
    print(s["workload"][:24], s["value"], s["ms_per_step"], s.get("step_frac"), s.get("step_frac_wall"), s.get("ms_per_step_ctypes_binding"))
PY
