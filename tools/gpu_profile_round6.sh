#!/bin/bash
# Round-6 evidence run on one MI355X box (via gpurun).  Summaries into gpurun_out/summ6/ (tools/collect_round_profiles.py 6
# copies them into profiles/ and rebuilds profiles/traffic_latest.json).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/summ6
mkdir -p $O
cd /tmp
# the default line twice: the driver's K (20 steps) and 100 steps -- burst and sustained figures in both
python3 $R/bench.py --steps 20 --warmup 5 > $O/r06_bench_cfg2_n1_driver_steps.json 2> $O/r06_bench_cfg2_n1_driver_steps.err
tail -1 $O/r06_bench_cfg2_n1_driver_steps.json | cut -c1-300
python3 $R/bench.py --steps 100 --warmup 20 > $O/r06_bench_cfg2_n1.json 2> $O/r06_bench_cfg2_n1.err
tail -1 $O/r06_bench_cfg2_n1.json | cut -c1-300
rocprofv3 --kernel-trace --stats -d $O/prof_cfg2 -o bench -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick --no-secondary > $O/r06_bench_cfg2_under_rocprof.json 2>/dev/null
python3 $R/tools/rocprof_summary.py $O/prof_cfg2 > $O/r06_bench_cfg2_kernel_stats.txt; rm -rf $O/prof_cfg2
grep "lsq::" $O/r06_bench_cfg2_kernel_stats.txt | cut -c1-200
for W in cfg1 cfg3 cfg4_shard cfg5 cfg5_bf16 tok tok_bf16 vit vit_bf16 cfg2_misaligned cfg5_channels_last cfg5_mixed_layout cfg5_bf16_channels_last cfg5_bf16_mixed_layout; do
  python3 $R/bench.py --workload $W --steps 200 --warmup 20 > $O/r06_bench_${W}_n1.json 2> $O/r06_bench_${W}_n1.err
  tail -1 $O/r06_bench_${W}_n1.json | cut -c1-200
done
for C in native native-inline c10d; do
  python3 $R/bench.py --workload cfg4_shard --assume-peers --collective $C --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick 2>/dev/null | grep "^{" > $O/r06_bench_cfg4_shard_collective_${C}.json
  tail -1 $O/r06_bench_cfg4_shard_collective_${C}.json | cut -c1-200
done
for W in cfg4_shard vit_bf16 cfg5_bf16 cfg5_bf16_mixed_layout; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$W -o bench -- python3 $R/bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
  python3 $R/tools/rocprof_summary.py $O/prof_$W > $O/r06_bench_${W}_kernel_stats.txt; rm -rf $O/prof_$W
  grep "lsq::" $O/r06_bench_${W}_kernel_stats.txt | cut -c1-200
done
cd $R
python3 bench.py --gpus 8 --backend gloo --single-device --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_bench_cfg4_8ranks_one_device_gloo.json 2> $O/r06_bench_8ranks.err
tail -1 $O/r06_bench_cfg4_8ranks_one_device_gloo.json | cut -c1-300
python3 bench.py --gpus 2 --backend gloo --single-device --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_bench_cfg2_2ranks_one_device_gloo.json 2> $O/r06_bench_2ranks.err
tail -1 $O/r06_bench_cfg2_2ranks_one_device_gloo.json | cut -c1-300
python3 tools/exp_bwd16_budget.py sweep > $O/r06_bwd16_grid_sweep.txt 2>&1; cat $O/r06_bwd16_grid_sweep.txt
python3 tools/exp_window_probe.py 2>/dev/null | head -2 > $O/r06_window_pattern_probe.txt
python3 tools/exp_host_breakdown.py > $O/r06_host_breakdown.txt 2>/dev/null
