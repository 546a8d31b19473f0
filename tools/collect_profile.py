#!/usr/bin/env python3
"""
Turns gpurun_out/<tag>/ (written by tools/profile.sh) into the committed evidence under profiles/:
  profiles/<name>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
  profiles/<name>_pmc.txt            per-kernel means of the PMC counters + derived figures
  profiles/traffic.json              HBM bytes per stage-kernel launch, keyed <solver>_<resolution>_<gpus>

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE and WRITE_SIZE are in KiB-like
units of 1024 B... FETCH_SIZE counts 64 B per 128-B request on gfx950 for streaming reads, so it is doubled;
WRITE_SIZE is exact.  traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024, averaged over the stage launches.
usage: tools/collect_profile.py <tag> <name> <key>
"""
import collections, csv, glob, json, os, shutil, sys

tag, name, key = sys.argv[1:4]
root = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
stats = glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, f"profiles/{name}_kernel_stats.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = []
fetch, write = [], []
for kname, counters in sorted(agg.items()):
    if "stage_kernel" not in kname and "wavespeed" not in kname and "::dg_" not in kname and "::srhd_" not in kname:
        continue
    lines.append(kname)
    mean = {c: sum(v) / len(v) for c, v in counters.items()}
    for c in sorted(mean):
        lines.append(f"   {c:24s} launches={len(counters[c]):3d} mean={mean[c]:.6g}")
    if "stage_kernel" in kname or "dg_advance" in kname or "dg_limit" in kname or "srhd_march" in kname:
        if "dg_limit" not in kname:  # traffic.json holds the dominant kernel's bytes
            fetch += counters.get("FETCH_SIZE", [])
            write += counters.get("WRITE_SIZE", [])
        if "SQ_ACTIVE_INST_VALU" in mean and "SQ_INSTS_VALU" in mean:
            lines.append(f"   derived: cycles per VALU instruction = {4 * mean['SQ_ACTIVE_INST_VALU'] / mean['SQ_INSTS_VALU']:.2f}")
            lines.append(f"   derived: VALU-active SIMD-cycles per launch = {4 * mean['SQ_ACTIVE_INST_VALU'] / 1024:.4g} per SIMD")
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            lines.append(f"   derived: HBM bytes per launch = {(2 * mean['FETCH_SIZE'] + mean['WRITE_SIZE']) * 1024:.6g}")
bench = os.path.join(root, "bench_line.json")
if os.path.exists(bench):
    lines.append("bench line of the --kernel-trace --stats run:")
    lines.append(open(bench).read().strip())
open(f"profiles/{name}_pmc.txt", "w").write("\n".join(lines) + "\n")
tpath = "profiles/traffic.json"
tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
if fetch and write:
    tj[key] = round((2 * sum(fetch) / len(fetch) + sum(write) / len(write)) * 1024)
json.dump(tj, open(tpath, "w"), indent=1, sort_keys=True)
print("\n".join(lines))
