#!/bin/bash
# GPU call M: full GPU suite after the host-layer split and the two moved thresholds; threshold sweep again; evidence run
mkdir -p gpurun_out/r03m
python -m pytest tests -m gpu -q -x > gpurun_out/r03m/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r03m/pytest.log
python tools/exp_policy_cliffs.py > gpurun_out/r03m/policy_cliffs.txt 2> gpurun_out/r03m/policy_cliffs.err; echo "cliffs rc=$?"
grep "threshold" gpurun_out/r03m/policy_cliffs.txt
