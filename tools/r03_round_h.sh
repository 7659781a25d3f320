#!/bin/bash
# Round 3, GPU call H: channel parameters requested before the row copies (prologue), timeline + bench + tests.
export TMPDIR=/tmp
O=gpurun_out/r03h
mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_dma_gpu.py tests/test_fuzz_gpu.py tests/test_policy_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
tail -4 $O/pytest.log | cut -c1-300
for W in cfg5_bf16 cfg5 tok_bf16 tok vit_bf16 vit cfg3; do
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
for W in cfg5_bf16 cfg5; do
  rocprofv3 --kernel-trace --stats -d $O/prof_${W} -o bench -- python3 bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
  python3 tools/rocprof_summary.py $O/prof_${W} > $O/kernel_stats_${W}.txt; rm -rf $O/prof_${W}
  grep "lsq::" $O/kernel_stats_${W}.txt | cut -c1-170
done
python3 tools/exp_timeline.py > $O/timeline.txt 2> $O/timeline.err
cat $O/timeline.txt | cut -c1-260; tail -3 $O/timeline.err
