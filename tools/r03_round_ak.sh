#!/bin/bash
# GPU call AK: segment kernels, one-group short walk (policy, 0) against the loop form (1): where does the short walk lose?
mkdir -p gpurun_out/r03ak
cd tools
SH="4x8x1048576@1 1x3x2000x2500@1 4x64x56x56@1 2x16x65536@1 6x32x8192@1 512x512x3x3@0 4096x4096@0 8x4194304@0 32000x4096@0"
for D in bf16 f32; do
python exp_knob_ab.py set_seg_no_up_front 0 1 $D $(for s in $SH; do echo f:$s; done) > ../gpurun_out/r03ak/fwd_$D.txt 2> ../gpurun_out/r03ak/err_f_$D.txt
python exp_knob_ab.py set_seg_no_up_front 0 1 $D $SH > ../gpurun_out/r03ak/bwd_$D.txt 2> ../gpurun_out/r03ak/err_b_$D.txt
done
cd ..
cat gpurun_out/r03ak/fwd_*.txt gpurun_out/r03ak/bwd_*.txt | cut -c1-200; tail -2 gpurun_out/r03ak/err*.txt
