"""Short per-kernel table from a rocprofv3 results db: calls, avg us, share (kernel names cut at the first '(')."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, count(*), sum(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id "
                  "group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for name, n, t in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    short = name.split("(")[0].replace("void ", "").replace("ppasr::", "")
    print(f"{short[:48]:48s} {n:5d} {t / n / 1e3:9.2f} us {100.0 * t / tot:5.1f} %")
