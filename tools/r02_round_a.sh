#!/bin/bash
# Round 2, GPU call A: VALU issue rates, GPU tests, bench lines of every workload, rocprofv3 kernel stats of the
# per-channel workloads, SQ counters of the backward kernels, cfg4 shard-step host-vs-GPU measurement.
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
tools/probes/valu_probe > $O/valu_probe.txt 2>&1
head -12 $O/valu_probe.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python tools/exp_shard_step.py > $O/shard_step.txt 2>&1
cat $O/shard_step.txt | cut -c1-400
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
cut -c1-300 $O/bench_cfg2.json
for W in cfg3 cfg5 cfg5_bf16 cfg1; do
  python bench.py --workload $W --steps 200 --warmup 20 > $O/bench_$W.json 2> $O/bench_$W.err
  python bench.py --workload $W --graph --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic > $O/bench_${W}_graph.json 2>> $O/bench_$W.err
  cut -c1-250 $O/bench_$W.json
done
for W in cfg3 cfg5 cfg5_bf16; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$W -o bench -- python3 bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > $O/bench_${W}_under_rocprof.json 2>/dev/null
  python3 tools/rocprof_summary.py $O/prof_$W > $O/kernel_stats_$W.txt
  head -4 $O/kernel_stats_$W.txt | cut -c1-200
  rm -rf $O/prof_$W
done
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS"
for W in cfg5_bf16 cfg5 cfg2; do
  i=0
  for SET in "$SQ1" "$SQ2" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $SET --kernel-trace -d $O/pmc_${W}_$i -o bench -- python3 bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
    python3 tools/rocprof_summary.py $O/pmc_${W}_$i --pmc | grep -E "^(SQ_|GRBM)" | grep "lsq::" | cut -c1-200 >> $O/sq_counters_$W.txt
    rm -rf $O/pmc_${W}_$i
  done
  cat $O/sq_counters_$W.txt | cut -c1-140
done
