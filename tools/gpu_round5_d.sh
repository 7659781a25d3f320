#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5e
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_shipped_binary_gpu.py tests/test_dma_gpu.py tests/test_ticket_gpu.py tests/test_policy_gpu.py tests/test_fuzz_gpu.py tests/test_owner_gpu.py -x -q > $O/tests_rg.txt 2>&1
tail -6 $O/tests_rg.txt | cut -c1-300
for W in vit_bf16 vit tok_bf16 tok; do
  timeout 300 python3 bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_$W.json 2> $O/bench_$W.err
  python3 - $W <<'PY'
import json, sys
w = sys.argv[1]
try:
    d = json.loads([l for l in open("gpurun_out/r5e/bench_%s.json" % w).read().splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    print(w, "value", d["value"], "ms", d["ms_per_step"], "bwd_ms", r["avg_launch_ms"], "bwd frac", r["frac"], "fwd_ms", r["fwd"]["avg_launch_ms"], "traffic x", r["traffic_over_algorithmic"], r["kernel"][:40])
except Exception as e:
    print(w, "failed", e)
PY
done
timeout 600 python3 tools/exp_knob_ab.py set_ww_cb 1 0 bf16 12608x768 3152x768 50432x768 8192x1024 16384x512 > $O/cb_ab_bf16.txt 2>&1
cat $O/cb_ab_bf16.txt | cut -c1-260
timeout 600 python3 tools/exp_knob_ab.py set_ww_cb 8 32 bf16 12608x768 50432x768 > $O/cb_ab_widths.txt 2>&1
cat $O/cb_ab_widths.txt | cut -c1-260
