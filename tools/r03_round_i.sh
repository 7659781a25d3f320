#!/bin/bash
# Round 3, GPU call I: row-group windows with staged lane parameters; per-SIMD timeline
export TMPDIR=/tmp
O=gpurun_out/r03i
mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_dma_gpu.py tests/test_fuzz_gpu.py tests/test_policy_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
tail -4 $O/pytest.log | cut -c1-300
for W in cfg5_bf16 cfg5 tok_bf16 tok vit_bf16 vit; do
  python3 bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic > $O/bench_${W}.json 2> $O/bench_${W}.err
  python3 - $O/bench_${W}.json $W <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-10s value %.1f  ms/step %.5f  bwd %.5f (med %.5f) frac %.4f  fwd %.5f frac %.4f  step_frac %.4f" % (
    sys.argv[2], d["value"], d["ms_per_step"], r["avg_launch_ms"], r["median_launch_ms"], r["frac"],
    r["fwd"]["avg_launch_ms"], r["fwd"]["frac"], r["step_frac"]))
PY
done
python3 tools/exp_timeline.py > $O/timeline.txt 2> $O/timeline.err
cat $O/timeline.txt | cut -c1-260; tail -3 $O/timeline.err
