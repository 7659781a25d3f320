#!/bin/bash
# round 5, first GPU call: the new tests, the default bench line (cfg4_shard_collective, ctypes A/B/A), the column-block probe
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_shipped_binary_gpu.py tests/test_rccl_world1_gpu.py tests/test_module_sync_gpu.py tests/test_owner_gpu.py -x -q > $O/tests_new.txt 2>&1
tail -25 $O/tests_new.txt
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.err
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5a/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "step_frac", d["roofline"]["step_frac"])
    print("ctypes", json.dumps(d["config"].get("python_ctypes_host_layer")))
    for s in d.get("secondary", []):
        print(json.dumps({k: v for k, v in s.items() if k not in ("what", "value_is", "per_op_ms_from")}))
except Exception as e:
    print("bench parse failed", e)
PY
timeout 900 python3 tools/exp_colblock_probe.py > $O/colblock_probe.txt 2>&1
cat $O/colblock_probe.txt | cut -c1-200
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1
tail -8 $O/tests_all.txt
