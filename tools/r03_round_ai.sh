#!/bin/bash
# GPU call AI: segment kernels compiled per walk kind (short: one group up front; long: the loop): tests, shape tables
mkdir -p gpurun_out/r03ai
python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_policy_gpu.py tests/test_foreach_gpu.py -q -x 2>&1 | tail -2
python tools/exp_activation_shapes.py > gpurun_out/r03ai/act.txt 2>/dev/null; grep "1048576\|2000, 2500\|3, 224" gpurun_out/r03ai/act.txt | cut -c1-200
python tools/exp_weight_shapes.py > gpurun_out/r03ai/weights.txt 2>/dev/null; cut -c1-170 gpurun_out/r03ai/weights.txt
