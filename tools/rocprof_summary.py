#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace sqlite database (rocpd) as a per-kernel table
(calls, total/avg ms, share) -- the same numbers `--stats` prints, kept as text under profiles/."""
import re
import sqlite3
import sys


def main(path, top=25, demangle=False):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(
        f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"
    ).fetchall()
    total = sum(r[2] for r in rows)
    print(f"# {path}\n# total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s}")
    for name, n, tot, mn, mx, vg, ag, sg, lds in rows[:top]:
        short = re.sub(r"\(.*$", "", name)[:70]
        if demangle:      # torch's element-wise kernels only differ in the functor buried deep in the mangled name: show its tail
            import subprocess
            full = subprocess.run(["c++filt", name.replace(".kd", "")], capture_output=True, text=True).stdout.strip()
            m = re.findall(r"(\w+(?:_kernel|Functor|_cuda|Ops|_impl)\w*)", full)
            short = (short[:28] + " " + ",".join(dict.fromkeys(m[-4:])))[:70]
        print(f"{short:70s} {n:6d} {tot/1e6:10.3f} {tot/n/1e3:9.1f} {mn/1e3:8.1f} {mx/1e3:8.1f} {100*tot/total:6.2f} {vg:5d} {ag:5d} {sg:5d} {lds:7d}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25, len(sys.argv) > 3 and sys.argv[3] == "--demangle")
