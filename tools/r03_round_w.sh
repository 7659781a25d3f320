#!/bin/bash
# GPU call W: forward without the channel table also for lanes of one / two channels (NCHW activations): A/B (knob 4 vs policy)
mkdir -p gpurun_out/r03w
cd tools
SH="f:256x2048x7x7@1 f:32x256x56x56@1 f:64x64x112x112@1 f:128x512x28x28@1 f:64x3x224x224@1 f:1x3x2000x2500@1 f:4x8x1048576@1 f:4x64x56x56@1"
python exp_knob_ab.py set_fwd_direct 4 0 bf16 $SH > ../gpurun_out/r03w/direct4_bf16.txt 2> ../gpurun_out/r03w/err1.txt
python exp_knob_ab.py set_fwd_direct 4 0 f32 $SH > ../gpurun_out/r03w/direct4_f32.txt 2> ../gpurun_out/r03w/err2.txt
cd ..
cat gpurun_out/r03w/direct4*.txt | cut -c1-250; tail -2 gpurun_out/r03w/err*.txt
