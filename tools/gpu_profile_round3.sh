#!/bin/bash
# Round-3 evidence run on one MI355X box (via gpurun): the default bench line (headline + secondary records, live PMC traffic,
# CPU baseline), rocprofv3 --kernel-trace --stats of the same command and of the secondary workloads, SQ counters of the 16-bit
# per-channel backward, reduction margins, multi-tensor table, K4 timeline, the 2-rank smoke.
# Summaries into gpurun_out/summ3/ (copied into profiles/ by tools/collect_round3.py).
export TMPDIR=/tmp
O=gpurun_out/summ3
mkdir -p $O
python3 bench.py --steps 100 --warmup 20 > $O/r03_bench_cfg2_n1.json 2> $O/r03_bench_cfg2_n1.err
tail -1 $O/r03_bench_cfg2_n1.json | cut -c1-300
rocprofv3 --kernel-trace --stats -d $O/prof_cfg2 -o bench -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick --no-secondary > $O/r03_bench_cfg2_under_rocprof.json 2>/dev/null
python3 tools/rocprof_summary.py $O/prof_cfg2 > $O/r03_bench_cfg2_kernel_stats.txt; rm -rf $O/prof_cfg2
grep "lsq::" $O/r03_bench_cfg2_kernel_stats.txt | cut -c1-200
for W in cfg1 cfg3 cfg5 cfg5_bf16 cfg5_axis0 tok tok_bf16 vit vit_bf16; do
  python3 bench.py --workload $W --steps 200 --warmup 20 > $O/r03_bench_${W}_n1.json 2> $O/r03_bench_${W}_n1.err
  tail -1 $O/r03_bench_${W}_n1.json | cut -c1-220
done
for W in cfg3 cfg5 cfg5_bf16 tok_bf16 vit_bf16; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$W -o bench -- python3 bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
  python3 tools/rocprof_summary.py $O/prof_$W > $O/r03_bench_${W}_kernel_stats.txt; rm -rf $O/prof_$W
  grep "lsq::" $O/r03_bench_${W}_kernel_stats.txt | cut -c1-200
done
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS"
for W in cfg5_bf16 cfg5; do
  i=0
  for SET in "$SQ1" "$SQ2" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $SET --kernel-trace -d $O/pmc_${W}_$i -o bench -- python3 bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
    python3 tools/rocprof_summary.py $O/pmc_${W}_$i --pmc | grep -E "^(SQ_|GRBM)" | grep "lsq::" | cut -c1-200 >> $O/r03_sq_counters_$W.txt
    rm -rf $O/pmc_${W}_$i
  done
done
grep "bwd_pc_kernel" $O/r03_sq_counters_cfg5_bf16.txt | cut -c1-150
python3 tools/exp_reduction_margin.py > $O/r03_reduction_margin.txt 2>/dev/null
python3 tools/exp_foreach.py > $O/r03_foreach_weights.txt 2>/dev/null
cat $O/r03_foreach_weights.txt | cut -c1-330
python3 tools/exp_timeline.py > $O/r03_k4_timeline.txt 2>/dev/null
python3 bench.py --gpus 2 --backend gloo --single-device --steps 20 --warmup 5 --no-cpu-baseline > $O/r03_bench_2ranks_one_device_gloo.json 2> $O/r03_bench_2ranks.err
tail -1 $O/r03_bench_2ranks_one_device_gloo.json | cut -c1-400
python3 __graft_entry__.py smoke > $O/r03_smoke.log 2>&1; tail -5 $O/r03_smoke.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r03_bench_driver_cmd.json 2> $O/r03_bench_driver_cmd.err; tail -4 $O/r03_bench_driver_cmd.err
