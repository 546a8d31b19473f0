#!/usr/bin/env python3
"""
Instruction budget of a marching kernel's loop from the compiler's assembly (no GPU needed):

    python tools/isa_budget.py [kernel-name-substring ...]   (default: the two RK2 stage kernels of cbdiso_2d)

compiles sfhip.hip to gfx950 assembly, finds the kernel, takes the longest backward-branch loop (the march) and
classifies its instructions: fp64 arithmetic (fma/mul/add), quarter-rate transcendental (rcp/rsq/sqrt), min/max,
compares and selects, DPP lane moves, plain moves, bit/integer ops, vector memory, scalar, waits.  Static counts
of the loop body including its rarely taken branches (sink interior, exact wavespeed); the PMC passes give the
dynamic count.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLASSES = [
    ("dpp lane move", lambda op, line: "dpp" in line or "row_" in line or "wave_sh" in line),
    ("fp64 fma/mul/add", lambda op, line: re.match(r"v_(fma|mul|add|fmac)_f64", op)),
    ("fp64 rcp/rsq/sqrt (quarter rate)", lambda op, line: re.match(r"v_(rcp|rsq|sqrt)_f64", op)),
    ("fp64 min/max", lambda op, line: re.match(r"v_(min|max)_f64", op)),
    ("fp64 other (ldexp, rndne, cvt, div_*, trig_preop, frexp)", lambda op, line: op.startswith("v_") and op.endswith("_f64")),
    ("compare", lambda op, line: op.startswith("v_cmp")),
    ("select (cndmask)", lambda op, line: op.startswith("v_cndmask")),
    ("move (v_mov, v_accvgpr)", lambda op, line: op.startswith("v_mov") or op.startswith("v_accvgpr")),
    ("readlane/readfirstlane/writelane", lambda op, line: "readlane" in op or "readfirstlane" in op or "writelane" in op),
    ("bit / integer VALU", lambda op, line: op.startswith("v_")),
    ("vector memory", lambda op, line: op.startswith(("global_", "buffer_", "flat_", "scratch_"))),
    ("LDS", lambda op, line: op.startswith("ds_")),
    ("wait / barrier / nop", lambda op, line: op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep"))),
    ("branch", lambda op, line: op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call"))),
    ("scalar ALU / load", lambda op, line: op.startswith("s_")),
]


def classify(op, line):
    for name, test in CLASSES:
        if test(op, line):
            return name
    return "other"


def main():
    patterns = sys.argv[1:] or ["iso_stage_kernelILi0ELb0ELb1", "iso_stage_kernelILi2ELb1ELb1"]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "sfhip.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17",
                        "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                        os.path.join(ROOT, "sailfish_amd", "csrc", "sfhip.hip"), "-o", asm],
                       check=True, capture_output=True)
        text = open(asm).read().splitlines()
    for pat in patterns:
        starts = [k for k, ln in enumerate(text) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", ln)]
        if not starts:
            print(f"no kernel matching {pat}")
            continue
        begin = starts[0]
        end = next(k for k in range(begin, len(text)) if text[k].strip().startswith(".Lfunc_end"))
        body = text[begin:end]
        labels = {m.group(1): k for k, ln in enumerate(body) if (m := re.match(r"^(\.LBB\w+):", ln))}
        loops = []
        for k, ln in enumerate(body):
            m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\w+)", ln)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                loops.append((k - labels[m.group(1)], labels[m.group(1)], k))
        if not loops:
            print(f"{pat}: no loop found")
            continue
        _, lo, hi = max(loops)
        counts = collections.Counter()
        valu = 0
        for ln in body[lo:hi + 1]:
            s = ln.split(";")[0].strip()
            if not s or s.endswith(":") or s.startswith("."):
                continue
            op = s.split()[0]
            c = classify(op, s)
            counts[c] += 1
            if op.startswith("v_"):
                valu += 1
        print(f"== {subprocess.run(['c++filt', body[0].split(':')[0]], capture_output=True, text=True).stdout.strip()[:100]}")
        print(f"   march loop: {sum(counts.values())} instructions, {valu} VALU (static, rarely taken branches included)")
        for name, _ in CLASSES + [("other", None)]:
            if counts[name]:
                print(f"   {counts[name]:5d}  {name}")


if __name__ == "__main__":
    main()
