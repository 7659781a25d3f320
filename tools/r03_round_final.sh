#!/bin/bash
# last GPU call of the round: the full GPU suite, then the evidence run
mkdir -p gpurun_out/r03final
python -m pytest tests -m gpu -q -x > gpurun_out/r03final/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r03final/pytest.log
bash tools/gpu_profile_round3.sh
