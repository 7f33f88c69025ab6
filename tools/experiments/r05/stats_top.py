"""Top kernels of a rocprofv3 --stats run: stats_top.py <dir> <steps>."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n = int(sys.argv[2])
for r in rows[:16]:
    print("%-64s calls/step %5.1f avg %8.1f us  %5.1f %%" % (r["Name"][:64], int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3,
                                                         100 * float(r["TotalDurationNs"]) / tot))
print("kernel time per step: %.3f ms" % (tot / n / 1e6))
