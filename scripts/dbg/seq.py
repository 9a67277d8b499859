# print the kernel sequence (name, grid, duration us) of one level-1 FlowStep of the backward sweep from a rocprofv3 kernel trace
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_cnet" in r["Kernel_Name"] and r["Kernel_Name"].rstrip().endswith("2>(glowhip::CnetArgs, glowhip::CnetGeo)")]
def show(i0, n=16):
    for r in rows[i0 - 6:i0 + n]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"  {r['Kernel_Name'][:70]:70s} grid {r.get('Grid_Size','?'):>8s} wg {r.get('Workgroup_Size','?'):>4s} {d:8.1f} us")
# level-3 launches come first in a sweep (top of the network), level-1 last: take the last, a middle and an early one of the last step
print("LEVEL 1"); show(idx[-2])
print("LEVEL 2"); show(idx[-2 - 40])
print("LEVEL 3"); show(idx[-2 - 80])
