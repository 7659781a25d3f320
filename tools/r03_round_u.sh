#!/bin/bash
# GPU call U: last-axis forward without the LDS channel table (lanes read their own scale / shift): parity, then A/B
mkdir -p gpurun_out/r03u
python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_dma_gpu.py tests/test_policy_gpu.py -q -x > gpurun_out/r03u/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r03u/pytest.log
cd tools
SH="f:8192x4096 f:12608x768 f:65536x1024 f:64x56x56x256@3 f:32x2048x4096@2 f:3152x768 f:128x768 f:200704x256 f:16384x8192"
python exp_knob_ab.py set_fwd_direct 1 2 bf16 $SH > ../gpurun_out/r03u/direct1_bf16.txt 2> ../gpurun_out/r03u/err1.txt
python exp_knob_ab.py set_fwd_direct 3 2 bf16 $SH > ../gpurun_out/r03u/direct3_bf16.txt 2> ../gpurun_out/r03u/err2.txt
python exp_knob_ab.py set_fwd_direct 1 2 f32 $SH > ../gpurun_out/r03u/direct1_f32.txt 2> ../gpurun_out/r03u/err3.txt
cd ..
cat gpurun_out/r03u/direct*.txt | cut -c1-250; tail -2 gpurun_out/r03u/err*.txt
