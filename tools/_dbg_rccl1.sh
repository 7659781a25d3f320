#!/bin/bash
rm -f /tmp/code.py; mkdir -p /tmp/dbg1; cd $GRAFT_REPO_ROOT
python3 - <<'PY' > /tmp/dbg1/rccl1_case.py
import sys
sys.path.insert(0, "tests")
import importlib.util, os
spec = importlib.util.spec_from_file_location("t", "tests/test_rccl_world1_gpu.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
print(m.CODE % {"root": os.getcwd()})
PY
MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 python3 -X faulthandler /tmp/dbg1/rccl1_case.py 2>&1 | grep -v "Warning\|warn\|super().__init__\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -40
