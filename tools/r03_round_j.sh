#!/bin/bash
# Round 3, GPU call J: smooth rows-per-workgroup rule (the 2^21 switch removed): tests + threshold sweep again + small shapes
export TMPDIR=/tmp
O=gpurun_out/r03j
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -4 $O/pytest.log | cut -c1-300
python3 tools/exp_policy_cliffs.py > $O/policy_cliffs.txt 2> $O/policy_cliffs.err
grep -c "CLIFF" $O/policy_cliffs.txt; grep -B4 -A3 "threshold" $O/policy_cliffs.txt | head -60 | cut -c1-220
python3 tools/exp_activation_shapes.py 2>&1 | grep -v amdgpu > $O/activation_shapes.txt
cat $O/activation_shapes.txt | cut -c1-250
