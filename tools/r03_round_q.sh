#!/bin/bash
# GPU call Q: policy tests again; small per-tensor config with the one-launch (ticket) backward on / off, eager
mkdir -p gpurun_out/r03q
python -m pytest tests/test_policy_gpu.py tests/test_foreach_gpu.py -q -x > gpurun_out/r03q/pytest.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r03q/pytest.log
for rep in 1 2; do
for T in 0 1; do
  for W in cfg1 cfg2; do
    TORCHLSQ_SINGLE_LAUNCH_BACKWARD=$T python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick --no-secondary > gpurun_out/r03q/bench_${W}_ticket$T.$rep.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open("gpurun_out/r03q/bench_${W}_ticket$T.$rep.json").read().strip().split("\n")[-1])
r=d["roofline"]
print("$W ticket=$T rep=$rep value %.1f ms_per_step %.5f bwd %.5f fwd %.5f wall_frac %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["fwd"]["avg_launch_ms"], r.get("step_frac_wall")))
PY
  done
done
done
