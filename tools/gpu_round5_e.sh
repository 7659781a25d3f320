#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O
cd /tmp
for CB in 1 16 8; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_cb$CB -o t -- python3 $GRAFT_REPO_ROOT/tools/exp_knob_ab.py set_ww_cb $CB $CB bf16 12608x768 > /dev/null 2>&1
  echo "## set_ww_cb $CB"
  python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $GRAFT_REPO_ROOT/$O/prof_cb$CB | grep "lsq::" | cut -c1-150
  rm -rf $GRAFT_REPO_ROOT/$O/prof_cb$CB
done
