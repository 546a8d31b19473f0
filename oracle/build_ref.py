#!/usr/bin/env python3
"""
Build recipe for ``oracle/_ref``: the reference's own C kernels, compiled in place.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sailfish_amd/`` imports or loads what
this script produces; it is used by ``tests/`` (to validate ``oracle/sf_oracle.c``
and the HIP kernels), by ``tests/golden/generate.py`` and by ``bench.py``'s
``cpu_baseline`` leg.

What it does (and does not do):

* The reference JIT-builds ``KERNEL_LIB_HEADER + <solver>.c`` through ``cffi``
  (``/root/reference/sailfish/kernel/library.py:26-87, 187-235``).  ``cffi`` is
  only a build driver there -- the product is a plain ``.so`` loaded with
  ``ctypes.CDLL`` (``library.py:228,232``).  Here the same translation unit is
  handed to ``gcc`` directly.
* No reference source text is copied into this repository: the header string is
  read out of ``library.py`` with ``ast`` (nothing is executed) and the solver
  ``.c`` file is read where it lies; both are piped to ``gcc -x c -`` on stdin.
  Only the resulting shared objects land in ``oracle/_ref/`` (git-ignored).
* Flags: ``-std=c99 -O3 -fPIC -shared -ffp-contract=off`` and no ``-march`` so
  that no FMA contraction can occur (the reference's default x86-64 build has
  none either).  ``omp`` flavour adds ``-fopenmp`` and
  ``-DEXEC_MODE=1`` exactly as ``sailfish/kernel/system.py:78-81`` does.

Run:  python oracle/build_ref.py            (no-op when /root/reference is absent)
"""
import ast
import os
import subprocess
import sys

REFERENCE = os.environ.get("SAILFISH_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
SOLVERS = ("cbdiso_2d", "cbdgam_2d", "srhd_2d", "cbdisodg_2d")
FLAVOURS = {"cpu": ["-DEXEC_MODE=0"], "omp": ["-DEXEC_MODE=1", "-fopenmp"]}
# srhd_2d takes PLM_THETA / MACH_CEILING as define_macros from its Solver (srhd_2d.py:225-233; defaults
# srhd_2d.py:66-67).  Extra builds with other values pin the oracle's run-time settings.
PROBED = ("cbdiso_2d", "cbdgam_2d")  # extra builds <solver>_probes.so: see ref_probes.c
MACRO_VARIANTS = {"srhd_2d": {"": [], "_theta2": ["-DPLM_THETA=2.0", "-DMACH_CEILING=100.0"]}}


def kernel_lib_header():
    path = os.path.join(REFERENCE, "sailfish", "kernel", "library.py")
    with open(path) as f:
        tree = ast.parse(f.read())
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(
            isinstance(t, ast.Name) and t.id == "KERNEL_LIB_HEADER" for t in node.targets
        ):
            return ast.literal_eval(node.value)
    raise RuntimeError("KERNEL_LIB_HEADER not found in reference library.py")


def build(verbose=True):
    if not os.path.isdir(REFERENCE):
        if verbose:
            print(f"[build_ref] {REFERENCE} not present; keeping any prebuilt oracle/_ref")
        return False
    os.makedirs(OUT, exist_ok=True)
    header = kernel_lib_header()
    for solver in SOLVERS:
        src_path = os.path.join(REFERENCE, "sailfish", "solvers", solver + ".c")
        with open(src_path) as f:
            # library.py:168 joins header and code with a single space
            code = f"{header} {f.read()}"
        for flavour, flags in FLAVOURS.items():
            for suffix, macros in MACRO_VARIANTS.get(solver, {"": []}).items():
                target = os.path.join(OUT, f"{solver}_{flavour}{suffix}.so")
                if os.path.exists(target) and os.path.getmtime(target) > max(
                    os.path.getmtime(src_path), os.path.getmtime(__file__)
                ):
                    continue
                # -w: srhd_2d.c calls printf/exit without including their headers (the reference's own
                # build only warns about it as well)
                cmd = ["gcc", "-std=c99", "-O3", "-fPIC", "-shared", "-ffp-contract=off", "-w",
                       *flags, *macros, "-x", "c", "-", "-o", target, "-lm"]
                if verbose:
                    print("[build_ref]", " ".join(cmd), "<", src_path)
                subprocess.run(cmd, input=code.encode(), check=True)
        if solver in PROBED:
            # the same translation unit with oracle/ref_probes.c behind it: exported loops over the file-local
            # helpers (slope, HLLE, primitive recovery) for tests/test_function_probes.py
            probes_path = os.path.join(HERE, "ref_probes.c")
            target = os.path.join(OUT, f"{solver}_probes.so")
            if not os.path.exists(target) or os.path.getmtime(target) <= max(
                os.path.getmtime(src_path), os.path.getmtime(probes_path), os.path.getmtime(__file__)
            ):
                with open(probes_path) as f:
                    probes = f.read()
                cmd = ["gcc", "-std=c99", "-O3", "-fPIC", "-shared", "-ffp-contract=off", "-w", "-DEXEC_MODE=0",
                       "-x", "c", "-", "-o", target, "-lm"]
                if verbose:
                    print("[build_ref]", " ".join(cmd), "<", src_path, "+", probes_path)
                subprocess.run(cmd, input=f"{code}\n{probes}".encode(), check=True)
    return True


if __name__ == "__main__":
    sys.exit(0 if build() or True else 1)
