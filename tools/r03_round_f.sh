#!/bin/bash
# Round 3, GPU call F: policy-pinning tests, the threshold sweep, the foreach table again (C++ host path), the K4 timeline.
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
timeout 1500 python -m pytest tests/test_policy_gpu.py tests/test_foreach_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python3 tools/exp_foreach.py > $O/foreach.txt 2> $O/foreach.err
cat $O/foreach.txt
python3 tools/exp_timeline.py > $O/timeline.txt 2> $O/timeline.err
cat $O/timeline.txt; tail -3 $O/timeline.err
python3 tools/exp_policy_cliffs.py > $O/policy_cliffs.txt 2> $O/policy_cliffs.err
cat $O/policy_cliffs.txt; tail -3 $O/policy_cliffs.err
