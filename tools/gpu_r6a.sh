#!/bin/bash
# round 6, first GPU pass: the communicator / bench verification changes + the default line
export TMPDIR=/tmp
O=gpurun_out/r6a
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_rccl_world1_gpu.py tests/test_bench_cli.py tests/test_policy_gpu.py tests/test_sharded_gpu.py tests/test_module_sync_gpu.py -m gpu -q > $O/tests_comm.txt 2>&1
tail -15 $O/tests_comm.txt | cut -c1-400
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'PY'
import json
lines=[l for l in open("gpurun_out/r6a/bench_default.json").read().splitlines()]
d=json.loads([l for l in lines if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "step_frac", d["roofline"]["step_frac"], "sustained", d["roofline"].get("sustained"))
for s in d.get("secondary", []):
    print(s["workload"], s.get("value"), s.get("ms_per_step"), s.get("fwd_ms"), s.get("bwd_ms"), s.get("error", ""), s.get("wall_over_solo_shard_step", ""), (s.get("collective_verified") or {}).get("ok"), (s.get("parity_vs_reference") or {}).get("ok"))
PY
