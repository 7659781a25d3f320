#!/bin/bash
# Round 3, GPU call D: the whole GPU suite after the production / tools split of the library, default bench line.
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python3 bench.py --steps 100 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err
python3 - $O/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["step_frac"], "secondary wall", d.get("secondary_wall_s"))
for r in d.get("secondary", []):
    print(r)
PY
python3 tools/exp_launch_geometry.py 2>&1 | tail -12
