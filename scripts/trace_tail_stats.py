"""Mean kernel durations over the LAST n dispatches of a rocprofv3 kernel trace (csv).  usage: trace_tail_stats.py <kernel_trace.csv> [n=400]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    a = acc[r["Kernel_Name"].split("(")[0][-48:]]
    a[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a[1] += 1
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = sum(v for v, _ in acc.values())
print(f"last {len(rows)} dispatches: {(t1 - t0) / 1e3:.0f} us wall, {busy:.0f} us inside kernels")
for k, (v, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:50s} n={c:4d} mean {v / c:7.1f} us")
