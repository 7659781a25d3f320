#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace --stats) of small / mid last-axis shapes, usual vs 768/1024-lane workgroups
export TMPDIR=/tmp
O=gpurun_out/r02z
mkdir -p $O
: > $O/ww_big_kernel_split.txt
for SPEC in "16,197,768 2 bfloat16" "16,197,768 2 float32" "64,197,768 2 bfloat16" "8,512,1280 2 bfloat16" "4096,1024 1 bfloat16"; do
  for K in 2 1; do
    set -- $SPEC
    rocprofv3 --kernel-trace --stats -d $O/p -o t -- python3 tools/exp_one_shape.py $1 $2 $3 $K > /dev/null 2>&1
    echo "== $SPEC ww_big=$K" >> $O/ww_big_kernel_split.txt
    python3 tools/rocprof_summary.py $O/p | grep -E "lsq::" | cut -c1-200 >> $O/ww_big_kernel_split.txt
    rm -rf $O/p
  done
done
cat $O/ww_big_kernel_split.txt
