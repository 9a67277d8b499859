"""Summarise the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes collected by scripts/prof_pmc.sh into
profiles/r01_pmc_hbm_traffic.txt and profiles/pmc_traffic.json (HBM bytes per launch of the dominant kernel)."""
import csv, collections, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def load(counter):
    rows = list(csv.DictReader(open(os.path.join(root, "gpurun_out", f"pmc_{counter}", "pmc_counter_collection.csv"))))
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]) if "Grid_Size" in r else 0)
        agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
    return agg
fetch, write = load("FETCH_SIZE"), load("WRITE_SIZE")
lines = []
for name, agg in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
    lines.append(f"== {name} (KB per dispatch, rocprofv3 --pmc {name} --kernel-trace; raw counter, uncorrected) ==")
    for (k, g), (tot, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        lines.append(f"{k[-60:]:60s} grid_threads={g:8d} dispatches={n:4d} avg_KB={tot / n:12.1f}")
open(os.path.join(root, "profiles", "r01_pmc_hbm_traffic.txt"), "w").write("\n".join(lines) + "\n")
# dominant kernel: k_gemm_sh, all launches (three grid sizes in equal numbers)
def per_launch(agg, pat):
    sel = {k: v for k, v in agg.items() if pat in k[0]}
    tot = sum(v[0] for v in sel.values()); n = sum(v[1] for v in sel.values())
    return tot / n * 1024.0, {str(k[1]): v[0] / v[1] for k, v in sel.items()}
f, fd = per_launch(fetch, "k_f02_sh")
w, wd = per_launch(write, "k_f02_sh")
out = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
out.pop("k_gemm_sh_hbm_bytes_per_launch", None); out.pop("k_gemm_sh_detail", None)
out["k_f02_sh_hbm_bytes_per_launch"] = int(2 * f + w)
out["k_f02_sh_detail"] = {
    "fetch_bytes_corrected_x2": int(2 * f), "write_bytes": int(w),
    "FETCH_SIZE_KB_raw_by_grid_threads": fd, "WRITE_SIZE_KB_by_grid_threads": wd,
    "algorithmic_bytes_avg": int((4 * 512 + 4 * 9) * 64 * (1024 + 256) / 2),   # write h2 + read z1 (6 / 12 channels), levels 1 and 2
    "method": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `bench.py --steps 2 --warmup 1` (scripts/prof_pmc.sh); "
              "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for 16 B/lane coalesced reads), "
              "KB->bytes x1024; average over all launches of k_f02_sh (levels 1 and 2 in equal numbers; the weight stream, 1.2 MB per workgroup, is served by L2), as bench.py's roofline does",
}
json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("k_f02_sh_hbm_bytes_per_launch", "k_f02_sh_detail")}, indent=1))
