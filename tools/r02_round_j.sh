#!/bin/bash
# Round 2, GPU call J: the whole GPU suite with the LDS-DMA defaults, activation-shape table, BASELINE config graph timing.
export TMPDIR=/tmp
O=gpurun_out/r02j
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
python tools/exp_activation_shapes.py 2>&1 | grep -v amdgpu > $O/activation_shapes.txt
cat $O/activation_shapes.txt
python tools/bench_configs.py --configs cfg1,cfg2,cfg3,cfg4s,cfg5,cfg5_bf16,cfg5_axis0 --graph-only 2>&1 | grep -v amdgpu > $O/graph_timing.txt
cut -c1-330 $O/graph_timing.txt
