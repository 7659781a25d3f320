#!/bin/bash
# Round-5 evidence run on one MI355X box (via gpurun).  Summaries into gpurun_out/summ5/ (tools/collect_round_profiles.py 5
# copies them into profiles/ and rebuilds profiles/traffic_latest.json).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/summ5
mkdir -p $O
cd /tmp
python3 $R/bench.py --steps 100 --warmup 20 > $O/r05_bench_cfg2_n1.json 2> $O/r05_bench_cfg2_n1.err
tail -1 $O/r05_bench_cfg2_n1.json | cut -c1-300
rocprofv3 --kernel-trace --stats -d $O/prof_cfg2 -o bench -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick --no-secondary > $O/r05_bench_cfg2_under_rocprof.json 2>/dev/null
python3 $R/tools/rocprof_summary.py $O/prof_cfg2 > $O/r05_bench_cfg2_kernel_stats.txt; rm -rf $O/prof_cfg2
grep "lsq::" $O/r05_bench_cfg2_kernel_stats.txt | cut -c1-200
for W in cfg1 cfg3 cfg4_shard cfg5 cfg5_bf16 tok tok_bf16 vit vit_bf16; do
  python3 $R/bench.py --workload $W --steps 200 --warmup 20 > $O/r05_bench_${W}_n1.json 2> $O/r05_bench_${W}_n1.err
  tail -1 $O/r05_bench_${W}_n1.json | cut -c1-200
done
for W in cfg1 cfg3 cfg4_shard vit_bf16; do
  python3 $R/bench.py --workload $W --steps 200 --warmup 20 --graph --no-cpu-baseline > $O/r05_bench_${W}_graph.json 2>/dev/null
  tail -1 $O/r05_bench_${W}_graph.json | cut -c1-200
done
# one rank's config-4 step WITH its collective (RCCL world of one told it has a peer), the three routes, and the kernel trace of the default one
for C in native native-inline c10d; do
  python3 $R/bench.py --workload cfg4_shard --assume-peers --collective $C --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick 2>/dev/null | grep "^{" > $O/r05_bench_cfg4_shard_collective_${C}.json
  tail -1 $O/r05_bench_cfg4_shard_collective_${C}.json | cut -c1-200
done
python3 $R/bench.py --workload cfg4_shard --assume-peers --collective native --graph --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick 2>/dev/null | grep "^{" > $O/r05_bench_cfg4_shard_collective_native_graph.json
tail -1 $O/r05_bench_cfg4_shard_collective_native_graph.json | cut -c1-200
for W in cfg4_shard vit_bf16 cfg5_bf16; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$W -o bench -- python3 $R/bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
  python3 $R/tools/rocprof_summary.py $O/prof_$W > $O/r05_bench_${W}_kernel_stats.txt; rm -rf $O/prof_$W
  grep "lsq::" $O/r05_bench_${W}_kernel_stats.txt | cut -c1-200
done
rocprofv3 --kernel-trace --stats -d $O/prof_col -o bench -- python3 $R/bench.py --workload cfg4_shard --assume-peers --collective native --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
python3 $R/tools/rocprof_summary.py $O/prof_col | head -12 > $O/r05_bench_cfg4_shard_collective_kernel_stats.txt; rm -rf $O/prof_col
cd $R
python3 bench.py --gpus 8 --backend gloo --single-device --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline > $O/r05_bench_cfg4_8ranks_one_device_gloo.json 2> $O/r05_bench_8ranks.err
tail -1 $O/r05_bench_cfg4_8ranks_one_device_gloo.json | cut -c1-300
python3 tools/exp_policy_audit.py 20 > $O/r05_policy_audit.txt 2>/dev/null; tail -3 $O/r05_policy_audit.txt
