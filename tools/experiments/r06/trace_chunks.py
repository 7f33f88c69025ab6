"""Per-chunk summary of a rocprofv3 kernel trace of tools/experiments/r06/stream_ab.py (SESS=1): launches, busy and wall time
per chunk over the last 40 chunks, per-kernel averages."""
import collections, csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(round(len(rows) / 240))
last = rows[-n * 40:]
t0 = int(last[0]['Start_Timestamp']); t1 = int(last[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last)
print(f"launches/chunk {n}  wall/chunk {(t1 - t0) / 40 / 1e3:.1f} us  busy/chunk {busy / 40 / 1e3:.1f} us")
agg = collections.defaultdict(lambda: [0, 0])
for r in last:
    k = r['Kernel_Name'].split('(')[0][:70]
    agg[k][0] += 1; agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:72s} n/chunk {c / 40:5.1f} avg us {t / c / 1e3:6.2f} tot/chunk {t / 40 / 1e3:7.1f}")
