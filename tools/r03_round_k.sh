#!/bin/bash
# GPU call K: policy tests after the smooth rule, row-group upper bound A/B
mkdir -p gpurun_out/r03k
python -m pytest tests/test_policy_gpu.py tests/test_abi.py -q -x > gpurun_out/r03k/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r03k/pytest.log
python tools/exp_ww_max.py > gpurun_out/r03k/ww_max_ab.txt 2> gpurun_out/r03k/ww_max_ab.err; echo "ww_max rc=$?"
cat gpurun_out/r03k/ww_max_ab.txt; tail -3 gpurun_out/r03k/ww_max_ab.err
