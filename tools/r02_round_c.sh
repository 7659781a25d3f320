#!/bin/bash
# Round 2, GPU call C: parity of the per-channel kernels after the geometry change (whole rounds, balanced rows) and the
# wave-wide-window last-axis backward; activation-shape table; BASELINE config graph timing.
export TMPDIR=/tmp
O=gpurun_out/r02c
mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_sharded_gpu.py -x -q > $O/pytest_parity.log 2>&1
tail -5 $O/pytest_parity.log
python tools/exp_activation_shapes.py > $O/activation_shapes.txt 2>&1
cat $O/activation_shapes.txt
python tools/bench_configs.py --configs cfg3,cfg5,cfg5_bf16,cfg5_axis0 --graph-only > $O/graph_timing.txt 2>&1
cut -c1-330 $O/graph_timing.txt
