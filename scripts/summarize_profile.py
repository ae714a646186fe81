#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV directory into profiles/<tag>_*.csv/.md (kept in git).

usage: scripts/summarize_profile.py gpurun_out/prof_r01/<host> r01
"""
import csv
import glob
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(repo, "profiles")
os.makedirs(out_dir, exist_ok=True)
# (a gpurun_out/ directory collects the files of every run that wrote there: the NEWEST pair is the run meant)
newest = lambda pat: max(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)   # noqa: E731
stats = newest("*_kernel_stats.csv")
trace = newest("*_kernel_trace.csv")
assert os.path.basename(stats).split("_")[0] == os.path.basename(trace).split("_")[0], (stats, trace)
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
groups = defaultdict(list)
meta = {}
for r in csv.DictReader(open(trace)):
    name = r["Kernel_Name"]
    if "dn::" not in name:
        continue
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    short = name.split("(")[0].replace("void ", "")
    groups[short].append(dur)
    meta[short] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                   r["Workgroup_Size_X"], r["Grid_Size_X"])
lines = [f"# rocprofv3 --kernel-trace --stats summary ({tag})", "",
         "Per-kernel launch durations of this repo's kernels (ns), from the kernel trace.  The fused network kernel's launches are",
         "listed per size (coarse / fine network of every configuration the command renders); bench.py's `roofline.kernel_ms` is the",
         "fine-net launch of the headline configuration (400 x 400 rays x 192 points: the ~22-25 ms cluster of the render precision's instance).", "",
         "| kernel | calls | avg ns | min ns | max ns | VGPR | AGPR | SGPR | LDS B | scratch B | wg | grid |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|"]
for k, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    m = meta[k]
    lines.append(f"| {k} | {len(v)} | {sum(v) / len(v):.0f} | {min(v)} | {max(v)} | " + " | ".join(m) + " |")
    if ("mlp_forward_kernel" in k or "mlp_forward48_kernel" in k) and len(v) > 2:
        # one kernel name serves several launch sizes (coarse / fine network of each configuration the command renders): cluster
        # the durations (a new cluster where the next duration is > 1.15x the previous one)
        clusters, cur = [], []
        for d in sorted(v):
            if cur and d > 1.15 * cur[-1]:
                clusters.append(cur); cur = []
            cur.append(d)
        clusters.append(cur)
        if len(clusters) > 1:
            for c in clusters:
                lines.append(f"| &nbsp;&nbsp;launches of ~{sum(c) / len(c) / 1e6:.2f} ms | {len(c)} | {sum(c) / len(c):.0f} | {min(c)} | {max(c)} | | | | | | | |")
open(os.path.join(out_dir, f"{tag}_kernel_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
