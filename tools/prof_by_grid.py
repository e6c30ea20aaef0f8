#!/usr/bin/env python
"""Steady-state breakdown of a rocprofv3 kernel trace (rocpd SQLite database) by (kernel, workgroups per launch): the
last `--steps` of `--of` traced iterations, per-step launch count, average duration and time share. Small grids with long
durations are launches that leave most of the chip idle (how the split-K candidates were found)."""
import argparse
import collections
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--of", type=int, required=True, help="iterations the trace covers (warm-up + timed)")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--top", type=int, default=24)
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    rows = cur.execute(f"select s.kernel_name, d.end - d.start, d.grid_size_x * d.grid_size_y * d.grid_size_z, "
                       f"d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z from {disp} d "
                       f"join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
    per = len(rows) // args.of
    rows = rows[-per * args.steps:]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, dur, g, w in rows:
        key = (name[:52], g // max(w, 1))
        agg[key][0] += 1
        agg[key][1] += dur / 1e3
    tot = sum(v[1] for v in agg.values())
    print(f"# {args.db}: {per} launches/step, {tot / args.steps / 1e3:.3f} ms of kernel time per step (one stream)")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f"{k[0]:54s} wgs={k[1]:7d} n/step={v[0] / args.steps:6.1f} avg={v[1] / v[0]:8.1f} us "
              f"per-step={v[1] / args.steps:9.1f} us {100 * v[1] / tot:5.1f} %")


if __name__ == "__main__":
    main()
