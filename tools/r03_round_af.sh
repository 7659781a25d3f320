#!/bin/bash
# GPU call AF: A/B of the direct allocator calls in the C++ binding (same box, interleaved)
mkdir -p gpurun_out/r03af
for rep in 1 2 3; do
for D in 1 0; do
for W in cfg3 cfg1; do
  TORCHLSQ_DIRECT_ALLOC=$D timeout 200 python bench.py --workload $W --steps 300 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > gpurun_out/r03af/bench_$W.$D.$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03af/bench_$W.$D.$rep.json").read().strip().split("\n")[-1])
print("$W direct=$D rep $rep value %.1f ms_per_step %.5f" % (d["value"], d["ms_per_step"]))
PY
done
done
done
timeout 600 python -m pytest tests/test_bench_cli.py -m gpu -q -x 2>&1 | tail -3
