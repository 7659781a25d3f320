#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5h
mkdir -p $O
cd /tmp
for R in native native-inline c10d; do
  python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4_shard --assume-peers --collective $R --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$R', d['config'].get('collective'), 'ms', d['ms_per_step'], 'fwd', d['roofline']['fwd']['avg_launch_ms'], 'bwd', d['roofline']['avg_launch_ms'], d['timed_blocks_ms_per_step'])"
done
python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4_shard --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('solo', 'ms', d['ms_per_step'], 'fwd', d['roofline']['fwd']['avg_launch_ms'], 'bwd', d['roofline']['avg_launch_ms'], d['timed_blocks_ms_per_step'])"
rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4_shard --assume-peers --collective native --steps 200 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-yardstick > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $O/prof | head -12 | cut -c1-170
python3 - <<'PY'
import glob, os, sqlite3
db = sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r5h/prof", "**", "*.db"), recursive=True))[-1]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select d.start, d.end, s.display_name, d.queue_id, d.stream_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
n = len(rows)
mid = rows[n // 2: n // 2 + 24]
t0 = mid[0][0]
for st, en, name, q, sid in mid:
    print("%9.2f %9.2f  q%s s%s  %s" % ((st - t0) / 1e3, (en - t0) / 1e3, q, sid, name[:70]))
PY
rm -rf $O/prof
