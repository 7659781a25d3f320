#!/bin/bash
# Round-2 evidence run on one MI355X box (via gpurun): every bench workload (JSON line with roofline, live PMC traffic,
# CPU baseline), rocprofv3 --kernel-trace --stats of the same commands, SQ counters of the 16-bit per-channel backward,
# the self-spawned 2-rank smoke.  Summaries into gpurun_out/summ2/ (copied to profiles/ by hand).
export TMPDIR=/tmp
O=gpurun_out/summ2
mkdir -p $O
python3 bench.py --steps 100 --warmup 20 > $O/r02_bench_cfg2_n1.json 2> $O/r02_bench_cfg2_n1.err
tail -1 $O/r02_bench_cfg2_n1.json | cut -c1-300
rocprofv3 --kernel-trace --stats -d $O/prof_cfg2 -o bench -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > $O/r02_bench_cfg2_under_rocprof.json 2>/dev/null
python3 tools/rocprof_summary.py $O/prof_cfg2 > $O/r02_bench_cfg2_kernel_stats.txt; rm -rf $O/prof_cfg2
head -5 $O/r02_bench_cfg2_kernel_stats.txt | cut -c1-200
for W in cfg3 cfg5 cfg5_bf16 cfg1 cfg5_axis0; do
  python3 bench.py --workload $W --steps 200 --warmup 20 > $O/r02_bench_${W}_n1.json 2> $O/r02_bench_${W}_n1.err
  tail -1 $O/r02_bench_${W}_n1.json | cut -c1-260
  python3 bench.py --workload $W --graph --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic > $O/r02_bench_${W}_graph.json 2>/dev/null
done
for W in cfg3 cfg5 cfg5_bf16; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$W -o bench -- python3 bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > $O/r02_bench_${W}_under_rocprof.json 2>/dev/null
  python3 tools/rocprof_summary.py $O/prof_$W > $O/r02_bench_${W}_kernel_stats.txt; rm -rf $O/prof_$W
  head -4 $O/r02_bench_${W}_kernel_stats.txt | cut -c1-200
done
# token-layout shapes (not BASELINE configs): bench line with live PMC traffic + kernel stats of the row-group-window kernels
if [ -z "$SKIP_TOKEN_SHAPES" ]; then
for W in tok tok_bf16 vit vit_bf16; do
  python3 bench.py --workload $W --steps 200 --warmup 20 > $O/r02_bench_${W}_n1.json 2> $O/r02_bench_${W}_n1.err
  tail -1 $O/r02_bench_${W}_n1.json | cut -c1-260
  rocprofv3 --kernel-trace --stats -d $O/prof_$W -o bench -- python3 bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > $O/r02_bench_${W}_under_rocprof.json 2>/dev/null
  python3 tools/rocprof_summary.py $O/prof_$W > $O/r02_bench_${W}_kernel_stats.txt; rm -rf $O/prof_$W
  head -5 $O/r02_bench_${W}_kernel_stats.txt | cut -c1-200
done
fi
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS"
for W in cfg5_bf16 cfg5; do
  i=0
  for SET in "$SQ1" "$SQ2" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $SET --kernel-trace -d $O/pmc_${W}_$i -o bench -- python3 bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
    python3 tools/rocprof_summary.py $O/pmc_${W}_$i --pmc | grep -E "^(SQ_|GRBM)" | grep "lsq::" | cut -c1-200 >> $O/r02_sq_counters_after_$W.txt
    rm -rf $O/pmc_${W}_$i
  done
done
grep "bwd_pc_kernel" $O/r02_sq_counters_after_cfg5_bf16.txt | cut -c1-150
python3 bench.py --gpus 2 --backend gloo --single-device --workload cfg4 --steps 50 --warmup 10 > $O/r02_bench_cfg4_2ranks_one_device_gloo.json 2> $O/r02_bench_cfg4_2ranks.err
tail -1 $O/r02_bench_cfg4_2ranks_one_device_gloo.json | cut -c1-300
python3 bench.py --gpus 4; echo "exit code of --gpus 4 on a 1-GPU box: $?" | tee $O/r02_bench_refuses_missing_gpus.txt
