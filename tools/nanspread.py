"""Which zones a NaN in one zone of the input reaches after one advance_rk stage: compiled oracle vs the HIP kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases, oracle
from sailfish_amd.kernel.library import Library

lib = Library(mode="hip"); orc = oracle.oracle()
for solver in ("iso", "gam"):
    for field in range(3 if solver == "iso" else 4):
        c = (cases.iso_case if solver == "iso" else cases.gam_case)(seed=5, ni=24, nj=30)
        prim = c["prim"].copy(); prim[2 + 12, 2 + 15, field] = np.nan
        name = "cbdiso_2d_advance_rk" if solver == "iso" else "cbdgam_2d_advance_rk"
        args = cases.iso_advance_args if solver == "iso" else cases.gam_advance_args
        expect = np.zeros_like(prim); orc.call(name, c["shape"], *args(c, c["cons"], prim, expect))
        wr = torch.zeros_like(torch.from_numpy(prim)).cuda()
        getattr(lib, name)[c["shape"]](*args(c, torch.from_numpy(c["cons"]).cuda(), torch.from_numpy(prim).cuda(), wr))
        got = wr.cpu().numpy()
        for label, a in (("oracle", expect), ("hip", got)):
            nan = np.isnan(a).any(-1)[2:-2, 2:-2]
            ii, jj = np.nonzero(nan)
            print(solver, "field", field, label, sorted(zip((ii - 12).tolist(), (jj - 15).tolist())))
        if solver == "iso":
            clean = np.zeros_like(prim); orc.call(name, c["shape"], *args(c, c["cons"], c["prim"], clean))
            for label, a in (("oracle", expect), ("hip", got)):
                bad = (np.abs(a - clean).max(-1) > 1e-6 * np.abs(clean).max())[2:-2, 2:-2]
                ii, jj = np.nonzero(bad)
                print(solver, "field", field, label, "changed", sorted(zip((ii - 12).tolist(), (jj - 15).tolist())))
                print("   values at the seeded zone", a[14, 17], "floor/ceiling", c.get("density_floor"), c.get("velocity_ceiling"))
            same = np.abs(got - expect).max(-1)[2:-2, 2:-2] <= 1e-12 * np.abs(expect).max()
            ii, jj = np.nonzero(~same); print("   hip != oracle at", sorted(zip((ii - 12).tolist(), (jj - 15).tolist())))
