#!/bin/bash
# round 6, second GPU pass: the whole GPU suite on the new build + the layout workloads (misaligned views, channels-last, mixed layout)
export TMPDIR=/tmp
O=gpurun_out/r6b
mkdir -p $O
for W in cfg2 cfg2_misaligned cfg5 cfg5_channels_last cfg5_mixed_layout cfg5_bf16 cfg5_bf16_channels_last cfg5_bf16_mixed_layout; do
  timeout 600 python3 bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-secondary > $O/bench_$W.json 2> $O/bench_$W.err
  python3 - "$O/bench_$W.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print("%-26s %8.2f GElem/s  %.5f ms  fwd %.5f bwd %.5f  step_frac %.4f  %s" % (d["config"]["workload"].split(":")[0], d["value"], d["ms_per_step"], r["fwd"]["avg_launch_ms"], r["avg_launch_ms"], r["step_frac"], d["config"]["layout"][:40]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
timeout 3000 python3 -m pytest tests -m gpu -q -x > $O/tests_all.txt 2>&1
tail -12 $O/tests_all.txt | cut -c1-400
