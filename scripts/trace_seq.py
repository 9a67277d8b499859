"""Debug: the dispatch sequence of one FlowStep's worth of kernels from a rocprofv3 kernel trace: name, grid, duration, gap to the
previous dispatch's end.  Usage: python scripts/trace_seq.py <kernel_trace.csv> <substring of the first kernel> [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat, cnt = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 12
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
i0 = idx[len(idx) * 3 // 4]                  # a late (steady-state) occurrence
for i in range(i0, min(i0 + cnt, len(rows))):
    r, q = rows[i], rows[i - 1]
    print(f"{r['Kernel_Name'][:70]:70s} grid {r.get('Grid_Size', '?'):>8s} wg {r.get('Workgroup_Size', '?'):>5s} "
          f"dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.2f} us  gap {(int(r['Start_Timestamp']) - int(q['End_Timestamp'])) / 1e3:7.2f} us")
