"""Idle time between consecutive kernel dispatches (all streams merged: GPU busy = union of the dispatch intervals) over
the MIDDLE third of a rocprofv3 --kernel-trace results db -- the un-instrumented timed region of a bench.py run -- plus
the largest gaps with the kernels on either side.  usage: python tools/rocpd_gaps.py x_results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
n = len(rows)
rows = rows[n // 3: 2 * n // 3]
short = lambda s: s.split("(")[0].replace("void ", "").replace("ppasr::", "")[:40]
t0, t1 = rows[0][1], max(r[2] for r in rows)
busy, cur_end, gaps = 0, rows[0][1], []
for name, st, en in rows:
    if st > cur_end:
        gaps.append((st - cur_end, prev, short(name)))
        busy += 0
        cur_start = st
    cur_end = max(cur_end, en)
    prev = short(name)
span = t1 - t0
idle = sum(g[0] for g in gaps)
print(f"dispatches {len(rows)}  span {span / 1e6:.3f} ms  idle {idle / 1e6:.3f} ms ({100.0 * idle / span:.1f} %)  gaps {len(gaps)}  mean gap {idle / max(len(gaps), 1) / 1e3:.2f} us")
by = {}
for g, a, b in gaps:
    k = (a, b)
    by.setdefault(k, [0, 0])
    by[k][0] += g
    by[k][1] += 1
for (a, b), (tot, cnt) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {tot / 1e3:9.1f} us total  {cnt:4d} x {tot / cnt / 1e3:7.2f} us   {a} -> {b}")
