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
        "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
        "group by kernel_name, counter_name order by kernel_name, counter_name"))
    if rows:
        print("\n# PMC (rocprofv3 --pmc, summed over XCDs/SEs): kernel | counter | dispatches | avg value per dispatch | avg ns")
        for r in rows:
            print(f"{r[0][:110]} | {r[1]} | {r[2]} | {r[3]:.0f} | {r[4]:.0f}")
except sqlite3.Error as e:
    print("# (no pmc table:", e, ")")
