#!/bin/bash
# GPU call R: ticket on/off by size; per-tensor ops on the per-channel tensors (how much is per-channel handling?)
mkdir -p gpurun_out/r03r
python tools/exp_ticket_sizes.py > gpurun_out/r03r/ticket_sizes.txt 2> gpurun_out/r03r/ticket_sizes.err; echo "rc=$?"
cat gpurun_out/r03r/ticket_sizes.txt | cut -c1-250
python tools/exp_pt_vs_pc.py > gpurun_out/r03r/pt_vs_pc.txt 2> gpurun_out/r03r/pt_vs_pc.err; echo "rc=$?"
cat gpurun_out/r03r/pt_vs_pc.txt | cut -c1-330
tail -3 gpurun_out/r03r/*.err
