#!/bin/bash
# GPU call L: policy tests with the 512 MB row-group bound; A/B of the band edges the threshold sweep flagged
mkdir -p gpurun_out/r03l
python -m pytest tests/test_policy_gpu.py tests/test_dma_gpu.py -q -x > gpurun_out/r03l/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r03l/pytest.log
cd tools
python exp_knob_ab.py set_ww_big 1 2 bf16 58000x768 65536x768 75366x768 87381x768 131072x768 49152x1024 65536x1024 98304x1024 24576x2048 32768x2048 49152x2048 > ../gpurun_out/r03l/ww_big_upper_bf16.txt 2> ../gpurun_out/r03l/err1.txt
python exp_knob_ab.py set_ww_big 1 2 f32 18568x768 21846x768 25122x768 28399x768 32768x768 16384x1024 24576x1024 8192x2048 12288x2048 > ../gpurun_out/r03l/ww_big_upper_f32.txt 2> ../gpurun_out/r03l/err2.txt
python exp_knob_ab.py force_ring 2 1 f32 46421x768 54614x768 62806x768 70998x768 87381x768 40960x1024 65536x1024 20480x2048 32768x2048 > ../gpurun_out/r03l/ring_upper_f32.txt 2> ../gpurun_out/r03l/err3.txt
cd ..
cat gpurun_out/r03l/ww_big_upper_bf16.txt gpurun_out/r03l/ww_big_upper_f32.txt gpurun_out/r03l/ring_upper_f32.txt
tail -2 gpurun_out/r03l/err*.txt
