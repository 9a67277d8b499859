"""Debug: idle time between consecutive dispatches of a rocprofv3 kernel trace (second half of the run = steady state): total kernel
time, total gap time, number of dispatches, gap histogram.  Usage: python scripts/trace_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
busy = gap = 0
hist = {1: 0, 2: 0, 5: 0, 10: 0, 50: 0, 1e9: 0}
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    if prev_end is not None and s > prev_end:
        g = (s - prev_end) / 1e3
        gap += s - prev_end
        for k in hist:
            if g <= k: hist[k] += 1; break
    prev_end = max(prev_end or 0, e)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{len(rows)} dispatches over {span/1e6:.2f} ms: kernels {busy/1e6:.2f} ms, gaps {gap/1e6:.2f} ms ({gap/span:.1%}); "
      f"gaps <=1us {hist[1]}, <=2 {hist[2]}, <=5 {hist[5]}, <=10 {hist[10]}, <=50 {hist[50]}, more {hist[1e9]}")
