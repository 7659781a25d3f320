#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite .db) into a small text table for profiles/.

    python tools/rocprof_summary.py <dir-or-db> [--pmc] > profiles/r01_....txt
Per kernel: calls, avg / min / max duration (us), total time share; with --pmc also the mean of every
collected counter per dispatch.
"""
import glob
import os
import sqlite3
import sys


def find_db(path):
    if os.path.isfile(path):
        return path
    c = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    if not c:
        raise SystemExit("no .db under " + path)
    return c[-1]


def main():
    path = sys.argv[1]
    want_pmc = "--pmc" in sys.argv
    db = sqlite3.connect(find_db(path))
    cur = db.cursor()
    rows = list(cur.execute(
        "select s.display_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start), "
        "max(d.grid_size_x), max(d.workgroup_size_x), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.display_name order by 6 desc"))
    total = sum(r[5] for r in rows) or 1
    print("# source: %s" % find_db(path))
    print("%-7s %-12s %-12s %-12s %-7s %-9s %-5s %-5s %-6s %s" % ("calls", "avg_us", "min_us", "max_us", "share%", "grid", "wg", "vgpr", "lds", "kernel"))
    for r in rows:
        name = r[0]
        if len(name) > 150:
            name = name[:150] + "..."
        print("%-7d %-12.2f %-12.2f %-12.2f %-7.2f %-9d %-5d %-5d %-6d %s" %
              (r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / total, r[6], r[7], r[8], r[10], name))
    if want_pmc:
        print("\n# counters: mean value per dispatch")
        q = ("select s.display_name, p.name, avg(e.value), count(*) from rocpd_pmc_event e "
             "join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id "
             "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.display_name, p.name order by 1, 2")
        try:
            for name, ctr, val, cnt in cur.execute(q):
                print("%-16s %-18.1f n=%-5d %s" % (ctr, val, cnt, name[:150]))
        except sqlite3.Error as e:
            print("pmc query failed:", e)
            for t in ("rocpd_pmc_event", "rocpd_info_pmc"):
                print(t, [c[1] for c in cur.execute("pragma table_info(%s)" % t)])


if __name__ == "__main__":
    main()
