#!/bin/bash
# Round 3, GPU call A: parity of the new arithmetic (v_med3 clamps, border = g * d, packed pairs), then the 16-bit
# per-channel backward A/B: 128-VGPR build (four workgroups per CU) against the unconstrained one (tools/_tune/liblsq_hip_w1.so).
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_dma_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1
tail -6 $O/pytest_parity.log
LIB=lsqfakequantize-pytorch_amd/torchlsq/liblsq_hip.so
bench_set() {   # $1 = tag
  for W in cfg5_bf16 cfg5 tok_bf16 vit_bf16; do
    python3 bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic > $O/bench_${W}_$1.json 2> $O/bench_${W}_$1.err
    python3 - $O/bench_${W}_$1.json $W $1 <<'EOF'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-10s %-4s value %.1f  ms/step %.5f  bwd %.5f (med %.5f) frac %.4f  fwd %.5f frac %.4f  step_frac %.4f" % (
    sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], r["avg_launch_ms"], r["median_launch_ms"], r["frac"],
    r["fwd"]["avg_launch_ms"], r["fwd"]["frac"], r["step_frac"]))
EOF
  done
  for W in cfg5_bf16 cfg5; do
    rocprofv3 --kernel-trace --stats -d $O/prof_${W}_$1 -o bench -- python3 bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
    python3 tools/rocprof_summary.py $O/prof_${W}_$1 > $O/kernel_stats_${W}_$1.txt; rm -rf $O/prof_${W}_$1
    grep "lsq::" $O/kernel_stats_${W}_$1.txt | cut -c1-170
  done
}
bench_set w4
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS"
counters() {   # $1 = tag
  for W in cfg5_bf16; do
    i=0
    for SET in "$SQ1" "$SQ2" "GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      rocprofv3 --pmc $SET --kernel-trace -d $O/pmc_${W}_$i -o bench -- python3 bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
      python3 tools/rocprof_summary.py $O/pmc_${W}_$i --pmc | grep -E "^(SQ_|GRBM)" | grep "lsq::" | cut -c1-200 >> $O/sq_counters_${W}_$1.txt
      rm -rf $O/pmc_${W}_$i
    done
  done
  grep "bwd_pc_kernel\|fwd_pc_kernel" $O/sq_counters_cfg5_bf16_$1.txt | cut -c1-150
}
counters w4
if [ -f tools/_tune/liblsq_hip_w1.so ]; then
  cp $LIB /tmp/liblsq_hip_w4.so
  cp tools/_tune/liblsq_hip_w1.so $LIB
  bench_set w1
  counters w1
  cp /tmp/liblsq_hip_w4.so $LIB
fi
