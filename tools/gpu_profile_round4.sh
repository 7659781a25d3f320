#!/bin/bash
# Round-4 evidence run on one MI355X box (via gpurun).  Summaries into gpurun_out/summ4/ (tools/collect_round_profiles.py 4
# copies them into profiles/).  Part 1: the default bench line + rocprofv3 kernel stats of the same command + the single-workload
# lines; part 2 (arg "exp"): the round's experiments.
export TMPDIR=/tmp
O=gpurun_out/summ4
mkdir -p $O
if [ "$1" != "exp" ]; then
python3 bench.py --steps 100 --warmup 20 > $O/r04_bench_cfg2_n1.json 2> $O/r04_bench_cfg2_n1.err
tail -1 $O/r04_bench_cfg2_n1.json | cut -c1-300
rocprofv3 --kernel-trace --stats -d $O/prof_cfg2 -o bench -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick --no-secondary > $O/r04_bench_cfg2_under_rocprof.json 2>/dev/null
python3 tools/rocprof_summary.py $O/prof_cfg2 > $O/r04_bench_cfg2_kernel_stats.txt; rm -rf $O/prof_cfg2
grep "lsq::" $O/r04_bench_cfg2_kernel_stats.txt | cut -c1-200
for W in cfg1 cfg3 cfg4_shard cfg5 cfg5_bf16; do
  python3 bench.py --workload $W --steps 200 --warmup 20 > $O/r04_bench_${W}_n1.json 2> $O/r04_bench_${W}_n1.err
  tail -1 $O/r04_bench_${W}_n1.json | cut -c1-220
done
for W in cfg1 cfg3 cfg4_shard; do
  python3 bench.py --workload $W --steps 200 --warmup 20 --graph --no-cpu-baseline > $O/r04_bench_${W}_graph.json 2>/dev/null
  tail -1 $O/r04_bench_${W}_graph.json | cut -c1-220
done
for W in cfg4_shard cfg5_bf16; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$W -o bench -- python3 bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
  python3 tools/rocprof_summary.py $O/prof_$W > $O/r04_bench_${W}_kernel_stats.txt; rm -rf $O/prof_$W
  grep "lsq::" $O/r04_bench_${W}_kernel_stats.txt | cut -c1-200
done
python3 bench.py --gpus 8 --backend gloo --single-device --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline > $O/r04_bench_cfg4_8ranks_one_device_gloo.json 2> $O/r04_bench_8ranks.err
tail -1 $O/r04_bench_cfg4_8ranks_one_device_gloo.json | cut -c1-400
else
# ---- experiments -------------------------------------------------------------------------------------------------------
python3 tools/exp_levels_only.py > $O/r04_levels_only.txt 2>/dev/null; cat $O/r04_levels_only.txt
for CTR in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CTR --kernel-trace -d $O/pmc_lv_$CTR -o lv -- python3 tools/exp_levels_only.py one cfg2 > /dev/null 2>&1
  python3 tools/rocprof_summary.py $O/pmc_lv_$CTR --pmc | grep -E "^$CTR" | grep "lsq::" | cut -c1-220 >> $O/r04_levels_only_pmc.txt
  rm -rf $O/pmc_lv_$CTR
done
cat $O/r04_levels_only_pmc.txt
python3 tools/exp_owner_probe.py > $O/r04_owner_pattern_probe.txt 2>/dev/null; cat $O/r04_owner_pattern_probe.txt
# (r04_finalize_lean_ab.txt: `exp_knob_ab.py set_fin_ch 64 0 ...` against the lean one-wave finalize kernel of commit 6576283,
#  removed after the measurement)
python3 tools/exp_owner_windows.py > $O/r04_owner_windows_ab.txt 2>/dev/null; cut -c1-200 $O/r04_owner_windows_ab.txt
for CTR in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CTR --kernel-trace -d $O/pmc_own_$CTR -o own -- python3 tools/exp_owner_probe.py one 1 16 49 1 > /dev/null 2>&1
  python3 tools/rocprof_summary.py $O/pmc_own_$CTR --pmc | grep -E "^$CTR" | grep "owner_add" | cut -c1-160 >> $O/r04_owner_pattern_probe_pmc.txt
  rm -rf $O/pmc_own_$CTR
done
cat $O/r04_owner_pattern_probe_pmc.txt
# SQ counters of the owner kernel next to the 256-lane-window kernel on config 5 (same driver, both kernels in one run)
: > $O/r04_sq_counters_owner_vs_windows.txt
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for DT in bf16 f32; do
    rocprofv3 --pmc $SET --kernel-trace -d $O/sq_${DT}_$i -o o -- python3 tools/exp_knob_ab.py set_own 1 2 $DT 256x2048x7x7@1 > /dev/null 2>&1
    echo "## $DT [256,2048,7,7] axis 1: bwd_pc_kernel<..., 512> = owner windows, <..., 256> = 256-lane windows" >> $O/r04_sq_counters_owner_vs_windows.txt
    python3 tools/rocprof_summary.py $O/sq_${DT}_$i --pmc | grep -E "^(SQ_|GRBM)" | grep "bwd_pc_kernel" | cut -c1-175 >> $O/r04_sq_counters_owner_vs_windows.txt
    rm -rf $O/sq_${DT}_$i
  done
done
python3 tools/exp_module_sync_cost.py 2>/dev/null > $O/r04_module_sync_cost.txt; cat $O/r04_module_sync_cost.txt | cut -c1-200
python3 tools/exp_timeline.py --build > /dev/null 2>&1      # the -DLSQ_TIMELINE experiment build is not shipped: made here (~1 min)
python3 tools/exp_timeline.py --own 2>/dev/null > $O/r04_owner_timeline.txt; grep -E "^##|busy span|per row|epilogue" $O/r04_owner_timeline.txt | cut -c1-220
fi
