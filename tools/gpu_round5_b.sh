#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p $O
timeout 600 python3 tools/exp_comm_cost.py > $O/comm_cost.txt 2>$O/comm_cost.err
cat $O/comm_cost.txt
timeout 1200 python3 -m pytest tests/test_shipped_binary_gpu.py tests/test_rccl_world1_gpu.py tests/test_module_sync_gpu.py tests/test_owner_gpu.py tests/test_policy_gate_gpu.py -q > $O/tests_new.txt 2>&1
tail -25 $O/tests_new.txt | cut -c1-250
