#!/usr/bin/env python3
"""
Turns gpurun_out/<tag>/ (written by tools/profile.sh) into the committed evidence under profiles/:
  profiles/<name>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
  profiles/<name>_pmc.txt            per-kernel means of the PMC counters + derived figures
  profiles/traffic.json              HBM bytes per stage-kernel launch, keyed <solver>_<resolution>_<gpus>
  profiles/valu.json                 SQ_INSTS_VALU per launch of the dominant kernel(s), cycles per instruction and the
                                     clock under the kernel (GRBM_GUI_ACTIVE / 8 XCDs / launch duration), same keys:
                                     what bench.py's roofline_valu is computed from

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE and WRITE_SIZE are in KiB-like
units of 1024 B... FETCH_SIZE counts 64 B per 128-B request on gfx950 for streaming reads, so it is doubled;
WRITE_SIZE is exact.  traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024, averaged over the stage launches.
usage: tools/collect_profile.py <tag> <name> <key>
"""
import collections, csv, glob, json, os, shutil, sys

tag, name, key = sys.argv[1:4]
root = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)


def newest(pattern):
    """gpurun MERGES a call's output into the local directory: a tag used twice holds both runs' files (named by
    process id), so only the most recent file of each pass directory is read."""
    by_dir = collections.defaultdict(list)
    for path in glob.glob(pattern, recursive=True):
        by_dir[os.path.dirname(path)].append(path)
    return [max(paths, key=os.path.getmtime) for paths in by_dir.values()]


stats = newest(os.path.join(root, "stats", "**", "*kernel_stats.csv"))[0]
shutil.copy(stats, f"profiles/{name}_kernel_stats.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
clocks = collections.defaultdict(list)  # GHz under each launch: GRBM_GUI_ACTIVE counts every XCD's cycles
for path in newest(os.path.join(root, "**", "*counter_collection.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            kname = row["Kernel_Name"].split("(")[0]
            agg[kname][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                ns = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                if ns > 0:
                    clocks[kname].append(float(row["Counter_Value"]) / 8.0 / ns)
lines = []
fetch, write = [], []
valu_insts, valu_cycles, valu_clock = [], [], []
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
        if "dg_limit" not in kname and "SQ_INSTS_VALU" in mean:
            valu_insts.append(mean["SQ_INSTS_VALU"])
            if "SQ_ACTIVE_INST_VALU" in mean:
                valu_cycles.append(4 * mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_INSTS_VALU"])
        if clocks.get(kname):
            # the first launches of the pass run on the ramp: the upper half is the sustained clock
            top = sorted(clocks[kname])[len(clocks[kname]) // 2:]
            ghz = sum(top) / len(top)
            lines.append(f"   derived: clock under the kernel = {ghz:.3f} GHz (GRBM_GUI_ACTIVE / 8 / duration, "
                         f"upper half of {len(clocks[kname])} launches)")
            if "dg_limit" not in kname and len(clocks[kname]) >= 10:  # (a variant launched once, on the ramp, says nothing)
                valu_clock.append(ghz)
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
vpath = "profiles/valu.json"
vj = json.load(open(vpath)) if os.path.exists(vpath) else {}
if valu_insts:
    vj[key] = dict(insts_per_launch=round(sum(valu_insts) / len(valu_insts)),
                   cycles_per_inst=round(sum(valu_cycles) / len(valu_cycles), 3) if valu_cycles else None,
                   clock_ghz=round(sum(valu_clock) / len(valu_clock), 3) if valu_clock else 2.1, source=name)
json.dump(vj, open(vpath, "w"), indent=1, sort_keys=True)
print("\n".join(lines))
