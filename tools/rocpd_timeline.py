"""Per-dispatch timeline of the LAST step in a rocprofv3 --kernel-trace results db: start offset, duration and the gap to
the previous kernel's end (negative = the dispatches overlap).  usage: python tools/rocpd_timeline.py x_results.db [n_last]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
rows = rows[-n_last:]
t0 = rows[0][1]
prev_end = None
tot_dur = 0
for name, st, en in rows:
    short = name.split("(")[0].replace("void ", "").replace("ppasr::", "")[:44]
    gap = "" if prev_end is None else f"{(st - prev_end) / 1e3:8.2f}"
    print(f"{short:44s} start {(st - t0) / 1e3:9.2f} us  dur {(en - st) / 1e3:8.2f} us  gap {gap}")
    prev_end = en
    tot_dur += en - st
print(f"span {(rows[-1][2] - t0) / 1e3:.2f} us, sum of durations {tot_dur / 1e3:.2f} us")
