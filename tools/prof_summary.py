#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`) into the
per-kernel table committed under profiles/: calls, total/avg/min/max duration, share of GPU kernel time."""
import re
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    rows = cur.execute(f"""select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start),
                                  max(d.end - d.start), min(d.start), max(d.end)
                           from {disp} d join {sym} s on d.kernel_id = s.id group by s.{name_col}
                           order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows)
    t0, t1 = min(r[5] for r in rows), max(r[6] for r in rows)
    print(f"# {path}")
    print(f"# kernels: {sum(r[1] for r in rows)} dispatches, {total / 1e6:.3f} ms busy, "
          f"{(t1 - t0) / 1e6:.3f} ms first-start to last-end ({100.0 * total / (t1 - t0):.1f} % busy)")
    print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'%':>6}  kernel")
    for name, n, tot, mn, mx, _, _ in rows[:top]:
        short = re.sub(r"\s+", " ", name)[:110]
        print(f"{n:8d} {tot / 1e6:10.3f} {tot / n / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} "
              f"{100.0 * tot / total:6.2f}  {short}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
