#!/bin/bash
export TMPDIR=/tmp
cd /tmp
for PICK in 1 0; do
  LSQ_COMM_PICK_STREAM=$PICK python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4_shard --assume-peers --collective native --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pick $PICK', 'ms', d['ms_per_step'], 'fwd', d['roofline']['fwd']['avg_launch_ms'], 'bwd', d['roofline']['avg_launch_ms'], d['timed_blocks_ms_per_step'])"
done
python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4_shard --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('solo', 'ms', d['ms_per_step'], 'fwd', d['roofline']['fwd']['avg_launch_ms'], 'bwd', d['roofline']['avg_launch_ms'], d['timed_blocks_ms_per_step'])"
cd $GRAFT_REPO_ROOT; timeout 600 python3 -m pytest tests/test_rccl_world1_gpu.py -x -q 2>&1 | tail -3
python3 tools/exp_comm_cost.py breakdown 2>&1 | grep "128x1024"
