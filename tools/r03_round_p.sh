#!/bin/bash
# GPU call P: short-row segment policy -- full GPU suite, multi-tensor table, small-config bench records
mkdir -p gpurun_out/r03p
python -m pytest tests -m gpu -q -x > gpurun_out/r03p/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r03p/pytest.log
python tools/exp_foreach.py > gpurun_out/r03p/foreach.txt 2> gpurun_out/r03p/foreach.err; echo "foreach rc=$?"
cut -c1-330 gpurun_out/r03p/foreach.txt
python tools/exp_weight_shapes.py > gpurun_out/r03p/weight_shapes.txt 2> gpurun_out/r03p/weight_shapes.err; echo "weights rc=$?"
cut -c1-250 gpurun_out/r03p/weight_shapes.txt
