#!/bin/bash
# GPU call T: row tiles dealt round-robin to the row splits (1) against contiguous runs (0, policy)
mkdir -p gpurun_out/r03t
cd tools
python exp_knob_ab.py set_row_interleave 1 0 bf16 256x2048x7x7@1 8192x4096 12608x768 32x256x56x56@1 64x64x112x112@1 65536x1024 64x56x56x256@3 4x8x1048576@1 f:256x2048x7x7@1 f:8192x4096 f:12608x768 f:65536x1024 > ../gpurun_out/r03t/interleave_bf16.txt 2> ../gpurun_out/r03t/err1.txt
python exp_knob_ab.py set_row_interleave 1 0 f32 256x2048x7x7@1 8192x4096 12608x768 32x256x56x56@1 64x64x112x112@1 65536x1024 64x56x56x256@3 f:256x2048x7x7@1 f:8192x4096 f:12608x768 > ../gpurun_out/r03t/interleave_f32.txt 2> ../gpurun_out/r03t/err2.txt
cd ..
cat gpurun_out/r03t/interleave_bf16.txt gpurun_out/r03t/interleave_f32.txt | cut -c1-260; tail -3 gpurun_out/r03t/err*.txt
