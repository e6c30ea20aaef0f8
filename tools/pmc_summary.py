#!/usr/bin/env python
"""Per-kernel PMC totals from a rocprofv3 --pmc rocpd database."""
import sqlite3, sys, collections
def main(path, filt=""):
    db = sqlite3.connect(path); cur = db.cursor()
    t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    ev = next(x for x in t if x.startswith("rocpd_pmc_event")); pmc = next(x for x in t if x.startswith("rocpd_info_pmc"))
    disp = next(x for x in t if x.startswith("rocpd_kernel_dispatch")); sym = next(x for x in t if x.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({ev})")]
    pcols = [r[1] for r in cur.execute(f"pragma table_info({pmc})")]
    scol = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in scol else "kernel_name"
    q = f"""select s.{name_col}, p.name, count(distinct d.id), sum(e.value) from {ev} e join {pmc} p on e.pmc_id = p.id
            join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id group by s.{name_col}, p.name"""
    res = collections.defaultdict(dict)
    for k, c, n, v in cur.execute(q):
        if filt in k: res[k[:70]][c] = (n, v)
    for k, d in res.items():
        print(k)
        for c, (n, v) in sorted(d.items()): print(f"   {c:32s} {v / n:16.1f} per dispatch ({n} dispatches)")
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
