#!/bin/bash
# GPU call Z: the one-launch per-channel backward (window closers fold their channels): ticket tests, parity, bench
mkdir -p gpurun_out/r03z
timeout 900 python -m pytest tests/test_ticket_gpu.py -q -x > gpurun_out/r03z/pytest_ticket.log 2>&1; echo "ticket tests rc=$?"
tail -4 gpurun_out/r03z/pytest_ticket.log
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_policy_gpu.py tests/test_dma_gpu.py -q -x > gpurun_out/r03z/pytest_parity.log 2>&1; echo "parity rc=$?"
tail -3 gpurun_out/r03z/pytest_parity.log
for W in cfg5_bf16 cfg5 cfg5_axis0; do
  for T in auto 0; do
  TORCHLSQ_SINGLE_LAUNCH_BACKWARD=$T timeout 300 python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > gpurun_out/r03z/bench_${W}_$T.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03z/bench_${W}_$T.json").read().strip().split("\n")[-1])
r=d["roofline"]
print("$W ticket=$T value %.1f ms_per_step %.5f bwd %.5f fwd %.5f bwd_frac %.4f step_frac %.4f wall %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["fwd"]["avg_launch_ms"], r["frac"], r["step_frac"], r.get("step_frac_wall")))
PY
  done
done
