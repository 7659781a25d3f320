#!/bin/bash
# Round 2, GPU call E: variant sweeps of the window-mode per-channel kernels on the new geometry (train and dx-only),
# kernel-level split (rocprofv3) of the last-axis shapes, compile tests.
export TMPDIR=/tmp
O=gpurun_out/r02e
mkdir -p $O
python -m pytest tests/test_compile_gpu.py -q > $O/pytest_compile.log 2>&1; tail -4 $O/pytest_compile.log
python tools/exp_pc_variants.py > $O/pc_variants_train.txt 2>&1; grep -v amdgpu $O/pc_variants_train.txt | cut -c1-1200
python tools/exp_pc_variants.py --eval > $O/pc_variants_eval.txt 2>&1; grep -v amdgpu $O/pc_variants_eval.txt | cut -c1-1200
for S in "64,197,768 2 float32" "64,197,768 2 bfloat16" "8192,4096 1 float32" "8192,4096 1 bfloat16" "64,56,56,256 3 bfloat16"; do
  T=$(echo $S | tr ' ,' '__')
  rocprofv3 --kernel-trace --stats -d $O/prof_$T -o x -- python3 tools/exp_one_shape.py $S > /dev/null 2>&1
  echo "== $S" >> $O/lastaxis_kernel_split.txt
  python3 tools/rocprof_summary.py $O/prof_$T | grep "lsq::" | cut -c1-190 >> $O/lastaxis_kernel_split.txt
  rm -rf $O/prof_$T
done
cat $O/lastaxis_kernel_split.txt
