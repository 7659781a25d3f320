#!/bin/bash
# only the token-layout workloads of tools/gpu_profile_round2.sh (bench line with live PMC traffic, rocprof kernel stats)
export TMPDIR=/tmp
O=gpurun_out/summ2
mkdir -p $O
for W in tok tok_bf16 vit vit_bf16; do
  python3 bench.py --workload $W --steps 200 --warmup 20 > $O/r02_bench_${W}_n1.json 2> $O/r02_bench_${W}_n1.err
  tail -1 $O/r02_bench_${W}_n1.json | cut -c1-400
  rocprofv3 --kernel-trace --stats -d $O/prof_$W -o bench -- python3 bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > $O/r02_bench_${W}_under_rocprof.json 2>/dev/null
  python3 tools/rocprof_summary.py $O/prof_$W > $O/r02_bench_${W}_kernel_stats.txt; rm -rf $O/prof_$W
  head -5 $O/r02_bench_${W}_kernel_stats.txt | cut -c1-200
done
