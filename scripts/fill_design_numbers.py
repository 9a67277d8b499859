#!/usr/bin/env python3
"""Fill the @@...@@ number slots of docs/DESIGN.md.in / docs/README.md.in from profiles/<tag>_bench.json and the rocprofv3 tables of the same
round: scripts/fill_design_numbers.py r06  (the .in files are the sources; DESIGN.md / README.md are generated and committed)."""
import csv
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
d = json.load(open(os.path.join(R, "profiles", f"{tag}_bench.json")))
sec, roof = d["secondary"], d["roofline"]
lv = roof["per_level"]


def n(v, nd=0):
    s = f"{v:,.{nd}f}".replace(",", " ")
    return s


def rocprof_avg(table, needle):
    with open(os.path.join(R, "profiles", table)) as f:
        for row in csv.DictReader(f):
            if needle in row["Name"]:
                return float(row["AverageNs"]) / 1e3
    return float("nan")


bd = d["breakdown_ms_per_step"]
fin = sum(v for k, v in bd.items() if k.startswith("cnet_finish"))
slots = {
    "B_VALUE": n(d["value"]), "B_MS": f"{d['ms_per_step']:.2f}", "B_FRAC": f"{roof['frac']:.3f}",
    "B_L1": f"{lv['C12_32x32']['frac']:.3f}", "B_L2": f"{lv['C24_16x16']['frac']:.3f}", "B_L3": f"{lv['C48_8x8']['frac']:.3f}",
    "L1_US": f"{lv['C12_32x32']['avg_launch_us']:.1f}", "L2_US": f"{lv['C24_16x16']['avg_launch_us']:.1f}",
    "L3_US": f"{lv['C48_8x8']['avg_launch_us']:.1f}",
    "L1_ROCPROF": f"{rocprof_avg(f'{tag}_fwd_B_kernel_stats.csv', 'k_cnet1w<512, 10, 4, 0, 1'):.1f}",
    "BC_VALUE": n(sec["B_forward_checked"]["value"]), "BC_MS": f"{sec['B_forward_checked']['ms_per_step']:.2f}",
    "BC_RATIO": f"{sec['B_forward_checked']['value'] / d['value']:.3f}",
    "D_VALUE": n(sec["D_forward"]["value"]), "E_VALUE": n(sec["E_forward"]["value"]), "EI_VALUE": n(sec["E_inverse"]["value"]),
    "T_VALUE": n(sec["B_train"]["value"]), "T_MS": f"{sec['B_train']['ms_per_step']:.2f}", "T_FRAC": f"{sec['B_train']['roofline']['frac']:.3f}",
    "CPU": f"{d['cpu_baseline']['value']:.2f} ({d['cpu_baseline']['cores']} cores)",
    "FIN_MS": f"{fin:.2f}",
    "BREAKDOWN": (f"`k_cnet1w` {bd['cnet_f0+f2+f4_C12_32x32']:.2f} (level 1), `k_cnet` {bd['cnet_f0+f2+f4_C24_16x16']:.2f} / "
                  f"{bd['cnet_f0+f2+f4_C48_8x8']:.2f} (levels 2 / 3), `k_cfinish` {bd['cnet_finish_C12_32x32']:.2f} / "
                  f"{bd['cnet_finish_C24_16x16']:.2f} / {bd['cnet_finish_C48_8x8']:.2f}, mixers "
                  f"{sum(v for k, v in bd.items() if k.startswith('chanmix')):.2f} ms; the pack's five launches ≈ 0.2 ms"),
}
for name in ("DESIGN.md", "README.md"):
    src = open(os.path.join(R, "docs", name + ".in")).read()
    for k, v in slots.items():
        src = src.replace(f"@@{k}@@", v)
    assert "@@" not in src, [l for l in src.splitlines() if "@@" in l][:3]
    open(os.path.join(R, name), "w").write(src)
    print(name, "written")
