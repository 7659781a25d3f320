#!/bin/bash
# GPU call O: segment walk for short channel rows (weights), A/B against the window kernels
mkdir -p gpurun_out/r03o
python tools/exp_seg_min.py --big > gpurun_out/r03o/seg_min_ab.txt 2> gpurun_out/r03o/seg_min_ab.err; echo "rc=$?"
cat gpurun_out/r03o/seg_min_ab.txt; tail -3 gpurun_out/r03o/seg_min_ab.err
