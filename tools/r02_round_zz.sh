#!/bin/bash
# Round 2, last refresh: activation-shape table, BASELINE config graph timing, last-axis kernel split with the final kernels.
export TMPDIR=/tmp
O=gpurun_out/r02zz
mkdir -p $O
python tools/exp_activation_shapes.py 2>&1 | grep -v amdgpu > $O/activation_shapes.txt
cat $O/activation_shapes.txt
python tools/bench_configs.py --configs cfg1,cfg2,cfg3,cfg4s,cfg5,cfg5_bf16,cfg5_axis0 --graph-only 2>&1 | grep -v amdgpu > $O/graph_timing.txt
cut -c1-330 $O/graph_timing.txt
: > $O/lastaxis_kernel_split.txt
for SPEC in "64,197,768 2 float32" "64,197,768 2 bfloat16" "8192,4096 1 float32" "8192,4096 1 bfloat16" "16,197,768 2 bfloat16"; do
  set -- $SPEC
  rocprofv3 --kernel-trace --stats -d $O/p -o t -- python3 tools/exp_one_shape.py $1 $2 $3 > /dev/null 2>&1
  echo "== $SPEC" >> $O/lastaxis_kernel_split.txt
  python3 tools/rocprof_summary.py $O/p | grep -E "lsq::" | cut -c1-200 >> $O/lastaxis_kernel_split.txt
  rm -rf $O/p
done
cat $O/lastaxis_kernel_split.txt
