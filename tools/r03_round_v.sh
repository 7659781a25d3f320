#!/bin/bash
# GPU call V: full GPU suite with the table-less last-axis forward; token-layout bench lines
mkdir -p gpurun_out/r03v
python -m pytest tests -m gpu -q -x > gpurun_out/r03v/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r03v/pytest.log
for W in tok tok_bf16 vit vit_bf16; do
  python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > gpurun_out/r03v/bench_$W.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03v/bench_$W.json").read().strip().split("\n")[-1])
r=d["roofline"]
print("$W value %.1f ms_per_step %.5f bwd %.5f fwd %.5f step_frac %.4f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["fwd"]["avg_launch_ms"], r["step_frac"]))
PY
done
