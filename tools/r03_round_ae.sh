#!/bin/bash
# GPU call AE: output buffers straight from the caching allocator in the C++ binding: small configs (host-bound), tests
mkdir -p gpurun_out/r03ae
for W in cfg3 cfg1; do
  for rep in 1 2 3; do
  python bench.py --workload $W --steps 300 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > gpurun_out/r03ae/bench_$W.$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03ae/bench_$W.$rep.json").read().strip().split("\n")[-1])
print("$W rep $rep value %.1f ms_per_step %.5f" % (d["value"], d["ms_per_step"]))
PY
  done
done
python tools/exp_foreach.py 2>/dev/null | grep native | cut -c1-330
python -m pytest tests -m gpu -q -x > gpurun_out/r03ae/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r03ae/pytest.log
