"""Debug: per (kernel, grid size) count / mean / min duration from a rocprofv3 kernel trace -- separates the levels of one kernel.
Usage: python scripts/trace_by_grid.py <kernel_trace.csv> [substring] [skip fraction of dispatches at the front, default 0.4]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.4
rows = rows[int(len(rows) * skip):]
g = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if pat not in n: continue
    n = re.sub(r"\(.*", "", n).replace("void glowhip::", "").replace("glowhip::", "")
    g[(n, "x".join(r[c] for c in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if c in r) or r.get("Grid_Size", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in g.values())
for (n, grid), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    q = sorted(v)
    print(f"{n[:60]:60s} grid {grid:>14s} x{len(v):5d} mean {sum(v)/len(v):8.2f} min {min(v):8.2f} us  share {sum(v)/tot:6.3f}"
          f"  sextiles 1/3/5: {q[len(q)//6]:.1f} {q[len(q)//2]:.1f} {q[len(q)*5//6]:.1f}")
