#!/bin/bash
# Round 3, GPU call B: the whole GPU suite on the new arithmetic, the default bench line with its secondary records,
# and the workgroups-per-CU sweep of the 16-bit ring backward now that its row costs ~16 instead of ~23 VALU per element.
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python3 bench.py --steps 100 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err
python3 - $O/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["step_frac"], "secondary wall", d.get("secondary_wall_s"))
for r in d.get("secondary", []):
    print(r)
PY
for W in cfg5_bf16 tok_bf16; do
  for BPC in 2 3 4 6 8 12 16; do
    V=$(( 1 + 768 + 8192 + BPC * 65536 ))
    python3 bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick --variant-bwd $V --host-binding ctypes > $O/sweep_${W}_$BPC.json 2>/dev/null
    python3 - $O/sweep_${W}_$BPC.json $W $BPC <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-10s ring, %2s wg/CU asked: bwd %.5f ms (median %.5f)  frac %.4f" % (sys.argv[2], sys.argv[3], r["avg_launch_ms"], r["median_launch_ms"], r["frac"]))
PY
  done
done
