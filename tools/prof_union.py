#!/usr/bin/env python
"""Multi-stream view of a rocprofv3 kernel trace (rocpd SQLite database) of graph-replayed steps: over the last `--steps`
of `--of` iterations, the wall span, the UNION of the kernels' busy intervals (time at least one kernel runs), the sum of
their durations (overlap factor) and the idle time — what stream-level scheduling could still win, as opposed to kernel
work."""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--of", type=int, required=True)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    rows = cur.execute(f"select start, end from {disp} order by start").fetchall()
    per = len(rows) // args.of
    rows = rows[-per * args.steps:]
    span = (max(e for _, e in rows) - rows[0][0]) / 1e6
    total = sum(e - s for s, e in rows) / 1e6
    union, cs, ce = 0, rows[0][0], rows[0][1]
    gaps = []
    for s, e in rows[1:]:
        if s > ce:
            union += ce - cs
            gaps.append((s - ce) / 1e3)
            cs, ce = s, e
        else:
            ce = max(ce, e)
    union = (union + ce - cs) / 1e6
    gaps.sort(reverse=True)
    n = args.steps
    print(f"# {per} launches/step; per step: span {span / n:.3f} ms, union busy {union / n:.3f} ms, sum of durations "
          f"{total / n:.3f} ms (overlap x{total / union:.2f}), idle {(span - union) / n:.3f} ms in {len(gaps) / n:.0f} gaps")
    print("# largest gaps (us):", [round(g, 1) for g in gaps[:12]])
    small = sum(g for g in gaps if g < 5.0)
    print(f"# gaps < 5 us: {small / n / 1e3:.3f} ms per step; >= 5 us: {(sum(gaps) - small) / n / 1e3:.3f} ms per step")


if __name__ == "__main__":
    main()
