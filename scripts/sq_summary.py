"""Summarise the rocprofv3 --pmc passes collected by scripts/prof_sq.sh (SQ / TCC / TCP / GRBM counters of bench.py's forward
step) into profiles/<tag>_pmc_sq.txt: per kernel (name, grid) the average counter values per dispatch and derived figures.
Usage: python scripts/sq_summary.py <gpurun_out subdir> <profiles tag>"""
import collections, csv, glob, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for path in sorted(glob.glob(os.path.join(root, "gpurun_out", src, "p*", "pmc_counter_collection.csv"))):
    seen = set()
    for r in csv.DictReader(open(path)):
        name = re.sub(r"^void glowhip::", "", r["Kernel_Name"].split("(")[0])
        key = (name[:48], int(r["Grid_Size"]), int(r["Workgroup_Size"]), r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
        a = agg[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
        did = (path, r["Dispatch_Id"])
        if did not in seen:
            seen.add(did)
            d = dur[key]; d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
lines = [f"# {tag}: rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (scripts/prof_sq.sh);",
         "# average per dispatch; SQ_* cycle counters are in quad-cycles summed over waves/SEs as rocprofv3 reports them;",
         "# us = dispatch duration under the profiler (counter passes run slower than un-profiled launches).",
         "# GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (a trivially busy kernel reads 8 x 2.4 GHz x duration)."]
try:
    head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"]).decode().strip()
    lines.append(f"# repo commit at collection time: {head}")
except Exception:
    pass
SKIP = ("k_conv_direct", "k_actnorm_init", "k_gemm_glds", "k_conv_tail", "k_conv_first", "__amd", "at::", "k_pack_scales")   # init pass / exact-fp32 comparison leg
order = [k for k in sorted(agg, key=lambda k: -dur[k][0]) if not any(s_ in k[0] for s_ in SKIP)]
for key in order[:int(os.environ.get("TOPN", "12"))]:
    c = {n: v[0] / v[1] for n, v in agg[key].items()}
    n_disp = max(v[1] for v in agg[key].values())
    us = dur[key][0] / max(dur[key][1], 1)
    lines.append("")
    lines.append(f"== {key[0]}  grid_threads={key[1]} wg={key[2]} vgpr={key[3]} agpr={key[4]} lds={key[5]}  dispatches/pass={n_disp}  avg_us={us:.1f}")
    for n in sorted(c):
        lines.append(f"   {n:36s} {c[n]:16.1f}")
    g = c.get("GRBM_GUI_ACTIVE")
    if g and us:
        lines.append(f"   -> effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration = {g / 8 / us / 1e3:.3f} GHz")
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs
        lines.append(f"   -> MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (g / 8 * 1024):.3f}")
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        parts = {n: c.get(n, 0) / w for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS")}
        lines.append("   -> of wave cycles: " + "  ".join(f"{n[3:]}={v:.3f}" for n, v in parts.items()))
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        lines.append(f"   -> LDS bank-conflict cycles / LDS active cycles = {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}")
    if "TCC_HIT_sum" in c:
        lines.append(f"   -> L2 hit rate = {c['TCC_HIT_sum'] / max(c['TCC_HIT_sum'] + c['TCC_MISS_sum'], 1):.3f}")
    if "TCP_TCC_READ_REQ_sum" in c and us:
        lines.append(f"   -> L1->L2 read requests/us = {c['TCP_TCC_READ_REQ_sum'] / us:.0f}  (x64 B = {c['TCP_TCC_READ_REQ_sum'] * 64 / us / 1e6:.2f} TB/s if 64-B requests)")
out = os.path.join(root, "profiles", f"{tag}_pmc_sq.txt")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:140]))
