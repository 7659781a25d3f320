#!/bin/bash
# GPU call AD: bench small configs after hoisting the op lookups; full GPU suite
mkdir -p gpurun_out/r03ad
for W in cfg3 cfg1; do
  for rep in 1 2; do
  python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > gpurun_out/r03ad/bench_$W.$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03ad/bench_$W.$rep.json").read().strip().split("\n")[-1])
r=d["roofline"]
print("$W rep $rep value %.1f ms_per_step %.5f bwd %.5f fwd %.5f step_frac %.4f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["fwd"]["avg_launch_ms"], r["step_frac"]))
PY
  done
done
python -m pytest tests -m gpu -q -x > gpurun_out/r03ad/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r03ad/pytest.log
