#!/usr/bin/env python3
"""
Register allocation of the marching kernels as the compiler reports it (no GPU needed):
    python tools/kernel_resources.py [pattern ...] [-DMACRO=value ...]
compiles sfhip.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and prints VGPRs, SGPRs, scratch
bytes per lane and waves per SIMD of every kernel whose demangled name contains one of the patterns
(default: the stage / march kernels).  Development tool; a kernel that starts to spill shows up here.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    flags = [a for a in sys.argv[1:] if a.startswith("-")]      # extra compiler flags, e.g. -DSFHIP_GAM_WAVES=3
    patterns = [a for a in sys.argv[1:] if not a.startswith("-")] or ["stage_kernel", "march_kernel", "dg_advance", "reduce_kernel"]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17",
               "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "sailfish_amd", "csrc", "sfhip.hip"),
               "-o", os.path.join(tmp, "x.o"), "-Rpass-analysis=kernel-resource-usage", *flags]
        text = subprocess.run(cmd, capture_output=True, text=True).stderr
    for block in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        mangled = block.split()[0]
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        if not any(p in name for p in patterns):
            continue

        def field(key):
            m = re.search(re.escape(key) + r": (\d+)", block)
            return m.group(1) if m else "?"

        short = name.split("(")[0].replace("void sf::", "")
        print(f"{short:48s} VGPR {field('VGPRs'):>4} AGPR {field('AGPRs'):>3} SGPR {field('SGPRs'):>4} "
              f"scratch {field('ScratchSize [bytes/lane]'):>4} waves/SIMD {field('Occupancy [waves/SIMD]')}")


if __name__ == "__main__":
    main()
