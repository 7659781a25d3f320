#!/bin/bash
# Run on the MI355X box (via gpurun): bench + rocprofv3 kernel stats + HBM counters, summaries into gpurun_out/summ/.
# usage: bash tools/gpu_profile_round.sh r01
R=${1:-r01}
OUT=gpurun_out/summ
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 100 --warmup 20 > $OUT/${R}_bench_n1.json 2> $OUT/${R}_bench_n1.err
tail -1 $OUT/${R}_bench_n1.json | cut -c1-400
# kernel stats of the very same command
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${R}_stats -o bench -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/${R}_bench_under_rocprof.json 2> /dev/null
python3 tools/rocprof_summary.py gpurun_out/prof_${R}_stats > $OUT/${R}_bench_kernel_stats.txt
head -5 $OUT/${R}_bench_kernel_stats.txt | cut -c1-200
# HBM traffic counters, each in its own pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d gpurun_out/prof_${R}_pmc_$C -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python3 tools/rocprof_summary.py gpurun_out/prof_${R}_pmc_$C --pmc | grep -E "^(FETCH_SIZE|WRITE_SIZE)" | grep "lsq::" > $OUT/${R}_bench_pmc_$C.txt
  rocprofv3 --pmc $C --kernel-trace -d gpurun_out/prof_${R}_cal_$C -o cal -- python3 tools/tune_stream.py --calibrate > /dev/null 2>&1
  python3 tools/rocprof_summary.py gpurun_out/prof_${R}_cal_$C --pmc | grep -E "^(FETCH_SIZE|WRITE_SIZE)" | grep -E "lsq::|probe_kernel" > $OUT/${R}_calibration_pmc_$C.txt
  cat $OUT/${R}_bench_pmc_$C.txt $OUT/${R}_calibration_pmc_$C.txt | cut -c1-160
done
rm -rf gpurun_out/prof_${R}_*/   # keep only the summaries (the .db files are tens of MB)
