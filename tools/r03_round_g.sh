#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03g
mkdir -p $O
python3 tools/exp_timeline.py > $O/timeline.txt 2> $O/timeline.err
cat $O/timeline.txt | cut -c1-260; tail -3 $O/timeline.err
timeout 1500 python -m pytest tests/test_policy_gpu.py -m gpu -q > $O/pytest.log 2>&1
tail -15 $O/pytest.log | cut -c1-300
