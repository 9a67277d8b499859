"""Summarise the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes collected by scripts/prof_pmc.sh into
profiles/<tag>_pmc_hbm_traffic.txt and profiles/pmc_traffic.json (HBM bytes per launch of the dominant kernel, k_cnet).
Usage: python scripts/pmc_summary.py <tag>     (reads gpurun_out/pmc_FETCH_SIZE, gpurun_out/pmc_WRITE_SIZE)"""
import csv, collections, json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
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
    for (k, g), (tot, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
        lines.append(f"{k[-60:]:60s} grid_threads={g:8d} dispatches={n:4d} avg_KB={tot / n:12.1f}")
open(os.path.join(root, "profiles", f"{tag}_pmc_hbm_traffic.txt"), "w").write("\n".join(lines) + "\n")
def per_launch(agg, pat):
    sel = {k: v for k, v in agg.items() if pat in k[0]}
    tot = sum(v[0] for v in sel.values()); n = sum(v[1] for v in sel.values())
    return tot / max(n, 1) * 1024.0, {f"{k[0][-28:]}@{k[1]}": v[0] / v[1] for k, v in sel.items()}
f, fd = per_launch(fetch, "k_cnet")
w, wd = per_launch(write, "k_cnet")
path = os.path.join(root, "profiles", "pmc_traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}
try:
    out["commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"]).decode().strip()
except Exception:
    out["commit"] = None
import hashlib
out["cnet_sh_sha16"] = hashlib.sha256(b"".join(open(os.path.join(root, "pytorch-glow_amd", "csrc", f), "rb").read()
                                               for f in ("cnet_sh.hip", "cnet1w_sh.hip"))).hexdigest()[:16]      # k_cnet + k_cnet1w
out["k_cnet_hbm_bytes_per_launch"] = int(2 * f + w)
out["k_cnet_detail"] = {
    "fetch_bytes_corrected_x2": int(2 * f), "write_bytes": int(w),
    "FETCH_SIZE_KB_raw_by_kernel@grid_threads": fd, "WRITE_SIZE_KB_by_kernel@grid_threads": wd,
    # config B, batch 64: read z1 (C/2 channels) + write the f.4 partial sums (Cout channels), fp32, per level, averaged over the
    # three levels' launches (equal numbers): pixels 65536 / 16384 / 4096, C = 12 / 24 / 48
    "algorithmic_bytes_avg": int(sum(4 * (c // 2 + c) * px for c, px in ((12, 65536), (24, 16384), (48, 4096))) / 3),
    "method": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline` "
              "(scripts/prof_pmc.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for 16 B/lane "
              "coalesced reads), KB->bytes x1024; average over ALL launches of k_cnet (three levels in equal numbers), as bench.py's "
              "roofline does.  The weight stream (1.4 MB per workgroup) is served by L2 / Infinity Cache; what reaches HBM beyond the "
              "algorithmic bytes is weights evicted between launches plus the halo rows.",
}
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("commit", "k_cnet_hbm_bytes_per_launch", "k_cnet_detail")}, indent=1)[:1500])
