#!/bin/bash
# GPU call AL: after the short-walk rule: full GPU suite, activation table
mkdir -p gpurun_out/r03al
python -m pytest tests -m gpu -q -x > gpurun_out/r03al/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r03al/pytest.log
python tools/exp_activation_shapes.py > gpurun_out/r03al/act.txt 2>/dev/null; grep "1048576\|2000, 2500" gpurun_out/r03al/act.txt | cut -c1-200
