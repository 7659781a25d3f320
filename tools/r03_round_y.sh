#!/bin/bash
# GPU call Y: default bench line (secondary records after the time-based warm-up), --graph mode, K4 timeline
mkdir -p gpurun_out/r03y
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03y/bench_driver_cmd.json 2> gpurun_out/r03y/bench_driver_cmd.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03y/bench_driver_cmd.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["step_frac"], d["roofline"].get("traffic_over_algorithmic"), d.get("secondary_wall_s"))
for s in d["secondary"]:
    print(s["workload"][:24], s.get("value"), s.get("ms_per_step"), s.get("step_frac"), s.get("step_frac_wall"), s.get("ms_per_step_ctypes_binding"), s.get("error"))
PY
for W in cfg5_bf16 cfg3; do
python bench.py --workload $W --graph --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > gpurun_out/r03y/bench_${W}_graph.json 2> gpurun_out/r03y/bench_${W}_graph.err; echo "graph rc=$?"
tail -1 gpurun_out/r03y/bench_${W}_graph.json | cut -c1-400
done
python tools/exp_timeline.py > gpurun_out/r03y/timeline.txt 2> gpurun_out/r03y/timeline.err; echo "timeline rc=$?"; head -12 gpurun_out/r03y/timeline.txt | cut -c1-200; tail -3 gpurun_out/r03y/timeline.err
