#!/bin/bash
# Round 3, GPU call E: multi-tensor (foreach) kernels: parity tests + timing table; eval-backward semantics tests.
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
timeout 1500 python -m pytest tests/test_foreach_gpu.py tests/test_qat_gpu.py tests/test_ticket_gpu.py tests/test_bench_cli.py -m gpu -x -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python3 tools/exp_foreach.py > $O/foreach.txt 2> $O/foreach.err
cat $O/foreach.txt
tail -3 $O/foreach.err
