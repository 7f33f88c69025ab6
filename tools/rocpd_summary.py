"""Dump the per-kernel summary (and PMC counters, if any) of a rocprofv3 rocpd sqlite db as text.
usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
print("# name | calls | total_us | avg_us | pct")
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{name} | {calls} | {total:.1f} | {avg:.2f} | {pct:.2f}")
try:
    rows = list(cur.execute(
        "select k.name, p.counter_name, count(*), sum(p.value), avg(p.value) from pmc_events p "
        "join kernels k on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name"))
    if rows:
        print("\n# PMC: kernel | counter | dispatches | sum | avg per dispatch")
        for r in rows:
            print(" | ".join(str(x) for x in r))
except sqlite3.Error as e:
    print("# (no pmc table:", e, ")")
