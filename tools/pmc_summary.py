#!/usr/bin/env python3
"""Per-kernel mean (per dispatch, summed over all XCD / SE instances) of hardware counters from a rocprofv3 --pmc sqlite
database (rocpd).   usage: pmc_summary.py results.db [COUNTER ...]   (no counter names: every counter in the database)"""
import re
import sqlite3
import sys


def main(path, counters):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pick = lambda pre: [t for t in tabs if t.startswith(pre)][0]
    ev, info, kd, ks = pick("rocpd_pmc_event"), pick("rocpd_info_pmc"), pick("rocpd_kernel_dispatch"), pick("rocpd_info_kernel_symbol")
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    key = "event_id" if "event_id" in cols else "id"
    try:
        rows = cur.execute(
            f"select i.name, s.kernel_name, count(*), avg(v), avg(d.end - d.start) from (select e.event_id as eid, e.pmc_id as pid, sum(e.value) as v "
            f"from {ev} e group by e.event_id, e.pmc_id) x join {info} i on x.pid = i.id join {kd} d on d.{key} = x.eid "
            f"join {ks} s on d.kernel_id = s.id group by i.name, s.kernel_name order by s.kernel_name, i.name"
        ).fetchall()
    except Exception as exc:  # schema drift: show what is there
        print("query failed:", exc, "\n", kd, cols, "\n", ev, [r[1] for r in cur.execute(f"pragma table_info({ev})")])
        return
    for name, kern, n, mean, dur in rows:
        if counters and name not in counters:
            continue
        short = re.sub(r"\(.*$", "", kern)[:70]
        print(f"{name:28s} {short:70s} n={n:3d} mean={mean:16.1f} avg_us={dur / 1e3:9.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
