"""
Test-side access to the CPU oracles (never imported by the package).

  ORACLE  -- oracle/libsf_oracle.so, our C restatement (symbols ``sfo_<kernel>``)
  REF     -- oracle/_ref/<solver>_{cpu,omp}.so, the reference's own C compiled in
             place by oracle/build_ref.py (symbols ``<kernel>``); absent on
             machines without /root/reference unless the prebuilt files travelled.

Signatures come from include/sfhip.h: the reference-named entry points there
have exactly the reference's argument lists, so one table serves all three
libraries.  Arrays are numpy float64.
"""
import ctypes
import os
import subprocess

import numpy as np

from sailfish_amd.kernel.parse_api import parse_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
with open(os.path.join(ROOT, "include", "sfhip.h")) as f:
    API = parse_api(f.read())

_CT = {"int": ctypes.c_int, "double": ctypes.c_double, "double*": ctypes.c_void_p}


class CLib:
    def __init__(self, path, prefix=""):
        self.path = path
        self.prefix = prefix
        self.dll = ctypes.CDLL(path)

    def has(self, name):
        return hasattr(self.dll, self.prefix + name)

    def call(self, name, shape, *args):
        sym = API[name]
        f = getattr(self.dll, self.prefix + name)
        f.restype = None
        f.argtypes = [_CT[a.dtype] for a in sym.args]
        full = list(shape) + list(args)
        assert len(full) == len(sym.args), f"{name}: {len(full)} args given, {len(sym.args)} expected"
        conv = []
        for v, a in zip(full, sym.args):
            if a.dtype == "double*":
                assert v.dtype == np.float64 and v.flags["C_CONTIGUOUS"]
                conv.append(v.ctypes.data)
            elif a.dtype == "int":
                conv.append(int(v))
            else:
                conv.append(float(v))
        f(*conv)


def build_oracle():
    """(Re)build oracle/libsf_oracle.so if missing or stale."""
    so = os.path.join(ORACLE_DIR, "libsf_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.startswith("sf_oracle") and f.endswith(".c")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src) for src in srcs):
        subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)
    return so


def oracle():
    return CLib(build_oracle(), prefix="sfo_")


def ref(solver, flavour="cpu"):
    """The compiled reference for `solver` ('cbdiso_2d' | 'cbdgam_2d'), or None."""
    path = os.path.join(ORACLE_DIR, "_ref", f"{solver}_{flavour}.so")
    if not os.path.exists(path) and os.path.isdir("/root/reference"):
        subprocess.run(["python3", os.path.join(ORACLE_DIR, "build_ref.py")], check=True)
    return CLib(path) if os.path.exists(path) else None
