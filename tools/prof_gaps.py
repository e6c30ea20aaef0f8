#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (rocpd SQLite database): for every dispatch the gap
between the end of the previous kernel and its own start, aggregated by the kernel that follows the gap. Gaps above
`--cut` microseconds (step boundaries, host stalls) are listed separately."""
import argparse
import re
import sqlite3
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--cut", type=float, default=1000.0)
    ap.add_argument("--last", type=int, default=0, help="only the last N dispatches (steady state)")
    ap.add_argument("--top", type=int, default=25)
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    rows = cur.execute(f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id "
                       f"order by d.start").fetchall()
    if args.last:
        rows = rows[-args.last:]
    busy = sum(e - s for _, s, e in rows)
    by_next, by_prev = defaultdict(lambda: [0, 0.0]), defaultdict(lambda: [0, 0.0])
    small, big, nbig = 0.0, 0.0, 0
    for (pn, ps, pe), (n, s, e) in zip(rows, rows[1:]):
        gap = (s - pe) / 1e3
        if gap > args.cut:
            big += gap; nbig += 1
            continue
        small += gap
        by_next[n][0] += 1; by_next[n][1] += gap
        by_prev[pn][0] += 1; by_prev[pn][1] += gap
    span = (rows[-1][2] - rows[0][1]) / 1e3
    print(f"# {args.db}: {len(rows)} dispatches, span {span / 1e3:.2f} ms, busy {busy / 1e6:.2f} ms, "
          f"gaps <= {args.cut:.0f} us: {small / 1e3:.2f} ms ({small / max(len(rows) - 1 - nbig, 1):.2f} us avg), "
          f"{nbig} larger gaps: {big / 1e3:.2f} ms")
    for title, table in (("kernel AFTER the gap", by_next), ("kernel BEFORE the gap", by_prev)):
        print(f"## by {title}\n{'count':>7} {'avg_gap_us':>11} {'total_ms':>9}  kernel")
        for name, (c, t) in sorted(table.items(), key=lambda kv: -kv[1][1])[:args.top]:
            short = re.sub(r"\s+", " ", name)[:100]
            print(f"{c:7d} {t / c:11.2f} {t / 1e3:9.3f}  {short}")


if __name__ == "__main__":
    main()
