#!/bin/bash
# GPU call Z2: kernel trace of config 5 bf16 with and without the one-launch per-channel backward
export TMPDIR=/tmp
mkdir -p gpurun_out/r03z2
for T in auto 0; do
  export TORCHLSQ_SINGLE_LAUNCH_BACKWARD=$T
  rocprofv3 --kernel-trace --stats -d gpurun_out/r03z2/prof_$T -o bench -- python3 bench.py --workload cfg5_bf16 --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
  python3 tools/rocprof_summary.py gpurun_out/r03z2/prof_$T > gpurun_out/r03z2/kernel_stats_$T.txt; rm -rf gpurun_out/r03z2/prof_$T
  echo "== ticket=$T"; grep "lsq::" gpurun_out/r03z2/kernel_stats_$T.txt | cut -c1-150
done
