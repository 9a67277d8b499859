"""Steady-state kernel statistics from a rocprofv3 kernel trace: everything up to the END of the data-dependent init pass (the
last k_actnorm_init dispatch and the full re-pack that follows it) is dropped, so the table shows the step being measured and
not the one-off set-up.  Usage: python scripts/trace_stats.py <t_kernel_trace.csv> <out_stats.csv>"""
import collections, csv, sys
src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_init = max((i for i, r in enumerate(rows) if "k_actnorm_init" in r["Kernel_Name"]), default=-1)
cut = last_init + 1
# the init pass ends with a full glowhip_plan_pack: skip through its last kernel (k_step_prepare_batched) as well
for i in range(cut, min(cut + 400, len(rows))):
    if "k_step_prepare_batched" in rows[i]["Kernel_Name"]:
        cut = i + 1
        break
agg = collections.OrderedDict()
for r in rows[cut:]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values()) or 1
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name, a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / tot, 2), a[2], a[3]])
print(f"{dst}: {len(rows) - cut} dispatches after the init pass ({cut} dropped), {len(agg)} kernels, {tot / 1e6:.2f} ms of kernel time")
