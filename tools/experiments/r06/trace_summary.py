"""Per-kernel table of a rocprofv3 --output-format csv kernel trace: calls, average and total time of the LAST `frac` of
the dispatches (python trace_summary.py trace.csv [frac])."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
last = rows[int(len(rows) * (1 - frac)):]
agg = collections.defaultdict(lambda: [0, 0])
for r in last:
    k = r['Kernel_Name'].split('(')[0][:72]
    agg[k][0] += 1; agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(v[1] for v in agg.values())
print(f"{len(last)} dispatches, busy {tot / 1e3:.1f} us, wall {(int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e3:.1f} us")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:74s} n {c:6d} avg us {t / c / 1e3:7.2f} share {t / tot:5.3f}")
