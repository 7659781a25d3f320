#!/bin/bash
# GPU call AC: segment walks of a few iterations issued as one group: tests, weight shapes, config 3
mkdir -p gpurun_out/r03ac
python -m pytest tests/test_foreach_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_policy_gpu.py -q -x > gpurun_out/r03ac/pytest.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r03ac/pytest.log
python tools/exp_weight_shapes.py > gpurun_out/r03ac/weight_shapes.txt 2>/dev/null; cut -c1-200 gpurun_out/r03ac/weight_shapes.txt
python tools/exp_foreach.py > gpurun_out/r03ac/foreach.txt 2>/dev/null; cut -c1-330 gpurun_out/r03ac/foreach.txt
for W in cfg3; do
  python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > gpurun_out/r03ac/bench_$W.json 2>/dev/null
  tail -1 gpurun_out/r03ac/bench_$W.json | cut -c1-330
done
