"""
Seeded kernel-level test problems: random-but-physical primitive data on small
non-square patches plus every scalar a kernel takes, for each branch combination
the kernels contain.
"""
import itertools

import numpy as np

NG = 2


def random_primitive(seed, ni, nj, nq):
    rng = np.random.default_rng(seed)
    p = np.empty((ni + 2 * NG, nj + 2 * NG, nq))
    p[..., 0] = rng.uniform(0.1, 2.0, p.shape[:2])
    p[..., 1] = rng.uniform(-2.0, 2.0, p.shape[:2])
    p[..., 2] = rng.uniform(-2.0, 2.0, p.shape[:2])
    if nq == 4:
        p[..., 3] = rng.uniform(0.1, 2.0, p.shape[:2])
    return p


def smooth_primitive(ni, nj, nq, extent):
    """Smooth data (so that slopes are mostly non-zero) on the given extent."""
    xl, xr, yl, yr = extent
    x = xl + (np.arange(-NG, ni + NG) + 0.5) * (xr - xl) / ni
    y = yl + (np.arange(-NG, nj + NG) + 0.5) * (yr - yl) / nj
    X, Y = np.meshgrid(x, y, indexing="ij")
    p = np.empty((ni + 2 * NG, nj + 2 * NG, nq))
    p[..., 0] = 1.0 + 0.3 * np.sin(1.3 * X) * np.cos(0.7 * Y)
    p[..., 1] = 0.5 * np.cos(0.9 * X + 0.2) - 0.2 * Y
    p[..., 2] = 0.4 * np.sin(0.5 * Y - 0.1) + 0.1 * X
    if nq == 4:
        p[..., 3] = 0.6 + 0.2 * np.cos(0.8 * X) * np.sin(1.1 * Y)
    return p


def mass(x, y, vx, vy, m, rsoft, rate, rsink, model):
    return (x, y, vx, vy, m, rsoft, rate, rsink, model)


NO_MASS = mass(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0)


def iso_case(seed=0, ni=32, nj=48, eos=2, nu=0.001, buffer_on=True, sink_model=2, a=0.0, masses=True,
             smooth=False, extent=(-1.6, 1.6, -2.4, 2.4), sink_rate=10.0, dt=1e-3, vmax=1e12, dfloor=1e-12):
    """Keyword dict for cbdiso_2d_advance_rk (minus the three arrays) plus the input arrays."""
    prim = smooth_primitive(ni, nj, 3, extent) if smooth else random_primitive(seed, ni, nj, 3)
    cons = np.zeros_like(prim)
    cons[..., 0] = prim[..., 0]
    cons[..., 1] = prim[..., 1] * prim[..., 0]
    cons[..., 2] = prim[..., 2] * prim[..., 0]
    rng = np.random.default_rng(seed + 1000)
    cons *= rng.uniform(0.9, 1.1, cons.shape)  # u^n differs from p2c(prim) in later stages
    if masses:
        m1 = mass(0.31, 0.12, -0.2, 0.6, 0.5, 0.05, sink_rate, 0.2, sink_model)
        m2 = mass(-0.29, -0.11, 0.2, -0.6, 0.5, 0.05, sink_rate, 0.2, sink_model)
    else:
        m1 = m2 = NO_MASS
    return dict(
        shape=(ni, nj), extent=extent, prim=prim, cons=cons,
        buffer=(0.8, m1[4] + m2[4], 100.0, max(abs(extent[1]), abs(extent[3])), 0.5, int(buffer_on)),
        m1=m1, m2=m2, eos=(1.3, 100.0, eos), nu=nu, a=a, dt=dt, vmax=vmax, dfloor=dfloor,
    )


def iso_advance_args(c, cons, rd, wr):
    return (*c["extent"], cons, rd, wr, *c["buffer"], *c["m1"], *c["m2"], *c["eos"],
            c["nu"], c["a"], c["dt"], c["vmax"], c["dfloor"])


def iso_wavespeed_args(c, prim, out):
    return (*c["extent"], *c["eos"], *c["m1"], *c["m2"], prim, out)


def iso_branch_matrix():
    """Every combination exercised by both the oracle-vs-reference and the HIP-vs-oracle tests."""
    combos = []
    for eos, nu, buf, model, a in itertools.product((1, 2), (0.0, 0.01), (0, 1), (0, 1, 2, 3), (0.0, 0.5)):
        combos.append(dict(eos=eos, nu=nu, buffer_on=bool(buf), sink_model=model, a=a))
    combos.append(dict(a=0.75, smooth=True))
    combos.append(dict(a=1.0 / 3.0, smooth=True))
    combos.append(dict(masses=False, eos=1, buffer_on=False, nu=0.0))
    combos.append(dict(sink_rate=-3.0, sink_model=1))  # negative rate: uniform source branch
    combos.append(dict(ni=5, nj=3))                    # tiny, ragged
    combos.append(dict(ni=1, nj=1))
    combos.append(dict(ni=67, nj=131, smooth=True))    # several strips / not multiples of anything
    # primitive recovery with its limits engaged (cbdiso_2d.c:268-283): with rho in [0.1, 2] and |v| <= 2 a
    # density floor of 0.6 catches about a quarter of the zones and a velocity ceiling of 0.9 about half
    combos.append(dict(dfloor=0.6))
    combos.append(dict(vmax=0.9))
    combos.append(dict(dfloor=0.6, vmax=0.9, a=0.5))
    combos.append(dict(dfloor=0.6, vmax=0.9, a=0.5, eos=1, sink_model=1, ni=67, nj=131))
    return combos


def iso_limits_engaged(c, out):
    """(zones whose density sits on the floor, zones with a velocity component on the ceiling) in an output array."""
    inner = out[NG:-NG, NG:-NG]
    return (int((inner[..., 0] == c["dfloor"]).sum()),
            int((np.abs(inner[..., 1:3]) == c["vmax"]).any(axis=-1).sum()))


def gam_case(seed=0, ni=32, nj=48, alpha=0.1, buffer_on=True, sink_model=2, a=0.0, masses=True, smooth=False,
             extent=(-1.6, 1.6, -2.4, 2.4), cooling=0.0, constant_softening=True, dt=1e-3, gamma=5.0 / 3.0,
             vmax=1e16, mach_ceiling=1e5, dfloor=1e-10, pfloor=1e-12):
    prim = smooth_primitive(ni, nj, 4, extent) if smooth else random_primitive(seed, ni, nj, 4)
    cons = np.zeros_like(prim)
    cons[..., 0] = prim[..., 0]
    cons[..., 1] = prim[..., 1] * prim[..., 0]
    cons[..., 2] = prim[..., 2] * prim[..., 0]
    cons[..., 3] = prim[..., 3] / (gamma - 1.0) + 0.5 * prim[..., 0] * (prim[..., 1] ** 2 + prim[..., 2] ** 2)
    rng = np.random.default_rng(seed + 1000)
    cons *= rng.uniform(0.95, 1.05, cons.shape)
    if masses:
        m1 = mass(0.31, 0.12, -0.2, 0.6, 0.5, 0.05, 10.0, 0.2, sink_model)
        m2 = mass(-0.29, -0.11, 0.2, -0.6, 0.5, 0.05, 10.0, 0.2, sink_model)
    else:
        m1 = m2 = NO_MASS
    return dict(
        shape=(ni, nj), extent=extent, prim=prim, cons=cons, gamma=gamma,
        buffer=(0.8, 0.05, m1[4] + m2[4], 1000.0, max(abs(extent[1]), abs(extent[3])), 0.5, int(buffer_on)),
        m1=m1, m2=m2, alpha=alpha, a=a, dt=dt, vmax=vmax, cooling=cooling, mach_ceiling=mach_ceiling,
        dfloor=dfloor, pfloor=pfloor, constant_softening=int(constant_softening),
    )


def gam_advance_args(c, cons, rd, wr):
    return (*c["extent"], cons, rd, wr, c["gamma"], *c["buffer"], *c["m1"], *c["m2"],
            c["alpha"], c["a"], c["dt"], c["vmax"], c["cooling"], c["mach_ceiling"], c["dfloor"], c["pfloor"],
            c["constant_softening"])


def gam_branch_matrix():
    combos = []
    for alpha, buf, model, a in itertools.product((0.0, 0.1), (0, 1), (0, 1, 2, 3), (0.0, 0.5)):
        combos.append(dict(alpha=alpha, buffer_on=bool(buf), sink_model=model, a=a))
    combos.append(dict(cooling=1e-2, a=0.5, smooth=True))
    combos.append(dict(constant_softening=False, a=0.5))
    combos.append(dict(masses=False, alpha=0.1, buffer_on=False))
    combos.append(dict(masses=False, alpha=0.0, buffer_on=False, smooth=True))
    combos.append(dict(ni=5, nj=3))
    combos.append(dict(ni=1, nj=1))
    combos.append(dict(ni=67, nj=131, smooth=True))
    # limits engaged (cbdgam_2d.c:298-345): pressure floor, velocity ceiling, the full reset of zones whose
    # density falls below the floor, and the Mach-ceiling floor of the cooling term (with and without cooling)
    combos.append(dict(pfloor=0.7))
    combos.append(dict(vmax=0.9, a=0.5))
    combos.append(dict(dfloor=0.6))
    combos.append(dict(dfloor=0.6, pfloor=0.7, vmax=0.9, a=0.5))
    combos.append(dict(mach_ceiling=0.8))
    combos.append(dict(mach_ceiling=0.8, cooling=1e-2, a=0.5, smooth=True))
    combos.append(dict(mach_ceiling=0.8, cooling=1e-2, dfloor=0.6, pfloor=0.7, vmax=0.9, ni=67, nj=131))
    return combos


def gam_limits_engaged(c, out):
    """(zones reset by the density floor, zones on the pressure floor, zones with a velocity on the ceiling)."""
    inner = out[NG:-NG, NG:-NG]
    reset = (inner[..., 0] == c["dfloor"]) & (inner[..., 1] == 0.0) & (inner[..., 2] == 0.0) & (inner[..., 3] == c["pfloor"])
    return (int(reset.sum()), int(((inner[..., 3] == c["pfloor"]) & ~reset).sum()),
            int((np.abs(inner[..., 1:3]) == c["vmax"]).any(axis=-1).sum()))


def component_relative_error(a, b, interior=True):
    """
    max over zones and fields of |a-b| / |b| where |b| is not negligible against the zone's largest field
    (a component that is a rounding-level remainder of the zone's scale has no relative accuracy to speak of:
    below 1e-6 of the zone's scale it is measured against that floor).
    """
    if interior:
        a = a[NG:-NG, NG:-NG]
        b = b[NG:-NG, NG:-NG]
    scale = np.abs(b).max(axis=-1, keepdims=True)
    den = np.maximum(np.abs(b), np.maximum(1e-6 * scale, 1e-300))
    return float((np.abs(a - b) / den).max())


def zone_relative_error(a, b, interior=True):
    """max over zones of |a-b|_inf / max(|b|_inf over the zone's fields, tiny)."""
    if interior:
        a = a[NG:-NG, NG:-NG]
        b = b[NG:-NG, NG:-NG]
    num = np.abs(a - b).max(axis=-1)
    den = np.maximum(np.abs(b).max(axis=-1), 1e-300)
    return float((num / den).max())


# ---------------------------------------------------------------------------
# srhd_2d: log-spherical mesh, (ni + 4, nj, 4) arrays with guards in i only
# ---------------------------------------------------------------------------
def srhd_case(seed=0, ni=24, nj=20, i0=0, zones_per_decade=40, adot=0.0, time=1.0, rk_param=0.0, jet=False,
              num_first_order_zones=0, polar_extent=np.pi, cold=False):
    """
    Seeded, physical srhd_2d state on rows [i0, i0 + ni) of a log mesh starting at r = 1:
    rho in [0.1, 2], gamma-beta u_r in [-1, 3], u_q in [-1, 1], p in [0.01, 2] modulated by a smooth
    profile (so the limiter sees both smooth and jagged data).  `cold` lowers the pressure by 1e-9
    so that conserved_to_primitive's Mach-ceiling floor engages.
    """
    rng = np.random.default_rng(seed)
    faces = np.array([10.0 ** (i / zones_per_decade) for i in range(i0, i0 + ni + 1)])
    shape = (ni + 4, nj, 4)
    ii = np.arange(ni + 4)[:, None] / (ni + 4)
    jj = (np.arange(nj)[None, :] + 0.5) / nj
    wave = 1.0 + 0.3 * np.sin(5.0 * ii) * np.cos(3.0 * jj)
    prim = np.empty(shape)
    prim[..., 0] = rng.uniform(0.1, 2.0, shape[:2]) * wave
    prim[..., 1] = rng.uniform(-1.0, 3.0, shape[:2]) * wave
    prim[..., 2] = rng.uniform(-1.0, 1.0, shape[:2])
    prim[..., 3] = rng.uniform(0.01, 2.0, shape[:2]) * wave * (1e-9 if cold else 1.0)
    a0 = 0.0 if adot else 1.0
    return dict(
        shape=(ni, nj), faces=faces, prim=prim, polar_extent=polar_extent, a0=a0, adot=adot, time=time,
        scale_factor=a0 + adot * time, rk_param=rk_param, dt=1e-3 * (a0 + adot * time),
        jet=(0.05, 8.0, 0.2, 5.0) if jet else (0.0, 0.0, 0.0, 0.0), nfo=num_first_order_zones,
    )


def srhd_conserved(lib, c, prim=None):
    """(ni + 4, nj, 4) conserved array whose interior is p2c(prim interior) by `lib`; guards zero."""
    prim = c["prim"] if prim is None else prim
    ni, nj = c["shape"]
    inner = np.ascontiguousarray(prim[2:-2])
    out = np.zeros_like(inner)
    lib.call("srhd_2d_primitive_to_conserved", c["shape"], c["faces"], inner, out, c["polar_extent"], c["scale_factor"])
    cons = np.zeros_like(prim)
    cons[2:-2] = out
    return cons


def srhd_advance_args(c, cons_rk, prim, cons_rd, cons_wr):
    return (c["faces"], cons_rk, prim, cons_rd, cons_wr, c["polar_extent"], c["a0"], c["adot"], c["time"],
            c["rk_param"], c["dt"], *c["jet"], c["nfo"])


def srhd_branch_matrix():
    return [
        dict(),
        dict(rk_param=0.5),
        dict(adot=0.7, time=1.3),
        dict(adot=0.7, time=1.3, rk_param=1.0 / 3.0, i0=7),
        dict(jet=True, time=1.2),
        dict(jet=True, time=9.0),                 # after the jet has switched off: innermost zones held
        dict(jet=True, time=1.2, adot=0.4, num_first_order_zones=5),
        dict(num_first_order_zones=100),          # first order everywhere
        dict(ni=3, nj=2),
        dict(ni=1, nj=1),
        dict(polar_extent=0.5 * np.pi, nj=33, ni=9),
        dict(ni=75, nj=130, num_first_order_zones=40, adot=0.1),   # several column strips and row chunks
        dict(ni=40, nj=61, jet=True, time=1.5),                    # a strip holding a single column
    ]


# ---------------------------------------------------------------------------
# cbdisodg_2d: weights arrays (ni + 2, nj + 2, 3, 3, 3), one guard zone per side
# ---------------------------------------------------------------------------
DG_USED = [(m, n) for m in range(3) for n in range(3) if m + n < 3]


def dg_weights(seed, ni, nj, extent, rough=0.0, moments=0.05):
    """
    Seeded modal weights: smooth zone averages (sigma > 0, momenta of either sign) with random
    higher moments of relative size `moments`; `rough` adds zone-to-zone noise to the averages so
    that the troubled-cell indicator and the limiter's branches fire.  The three unused slots of
    each field hold a sentinel so tests can see that they are carried along / left alone.
    """
    rng = np.random.default_rng(seed)
    xl, xr, yl, yr = extent
    x = xl + (np.arange(-1, ni + 1) + 0.5) * (xr - xl) / ni
    y = yl + (np.arange(-1, nj + 1) + 0.5) * (yr - yl) / nj
    X, Y = np.meshgrid(x, y, indexing="ij")
    w = np.full((ni + 2, nj + 2, 3, 3, 3), 77.0)
    avg = np.empty((ni + 2, nj + 2, 3))
    avg[..., 0] = 1.0 + 0.3 * np.sin(1.3 * X) * np.cos(0.7 * Y)
    avg[..., 1] = avg[..., 0] * (0.5 * np.cos(0.9 * X + 0.2) - 0.2 * Y)
    avg[..., 2] = avg[..., 0] * (0.4 * np.sin(0.5 * Y - 0.1) + 0.1 * X)
    if rough:
        avg *= 1.0 + rough * rng.uniform(-1.0, 1.0, avg.shape)
    for m, n in DG_USED:
        if m + n == 0:
            w[..., m, n] = avg
        else:
            w[..., m, n] = moments * rng.uniform(-1.0, 1.0, avg.shape) * np.abs(avg[..., :1])
    return w


def dg_case(seed=0, ni=20, nj=28, eos=2, nu=0.001, buffer_on=True, sink_model=2, rk_param=0.0, masses=True,
            extent=(-1.6, 1.6, -2.4, 2.4), sink_rate=10.0, dt=1e-3, rough=0.0, moments=0.05, big_slopes=False):
    w1 = dg_weights(seed, ni, nj, extent, rough, moments)
    w0 = dg_weights(seed + 500, ni, nj, extent, rough, moments)
    if big_slopes:  # slopes beyond +-1 so that the limiter's integer truncation returns non-zero values
        rng = np.random.default_rng(seed + 9)
        w1[..., 1, 0] = rng.uniform(-6.0, 6.0, w1.shape[:3])
        w1[..., 0, 1] = rng.uniform(-6.0, 6.0, w1.shape[:3])
        w1[..., 0, 0] += rng.integers(-4, 5, w1.shape[:3]) * np.sqrt(3.0) * 2.5
        w1[..., 0, 0, 0] = np.abs(w1[..., 0, 0, 0]) + 0.5
    if masses:
        m1 = mass(0.31, 0.12, -0.2, 0.6, 0.5, 0.05, sink_rate, 0.2, sink_model)
        m2 = mass(-0.29, -0.11, 0.2, -0.6, 0.5, 0.05, sink_rate, 0.2, sink_model)
    else:
        m1 = m2 = NO_MASS
    return dict(
        shape=(ni, nj), extent=extent, w0=w0, w1=w1,
        buffer=(0.8, m1[4] + m2[4], 100.0, max(abs(extent[1]), abs(extent[3])), 0.5, int(buffer_on)),
        m1=m1, m2=m2, eos=(1.3, 100.0, eos), nu=nu, rk_param=rk_param, dt=dt, vmax=1e12,
    )


def dg_advance_args(c, w0, w1, w2):
    return (*c["extent"], w0, w1, w2, *c["buffer"], *c["m1"], *c["m2"], *c["eos"], c["nu"], c["rk_param"], c["dt"],
            c["vmax"])


def dg_limit_args(c, w1, w2):
    return (*c["extent"], c["m1"][0], c["m1"][1], c["m2"][0], c["m2"][1], w1, w2)


def dg_wavespeed_args(c, w, out):
    return (*c["extent"], *c["eos"], *c["m1"], *c["m2"], c["vmax"], w, out)


def dg_branch_matrix():
    combos = []
    for eos, nu, buf, model, a in itertools.product((1, 2), (0.0, 0.01), (0, 1), (0, 1, 2, 3), (0.0, 0.5)):
        combos.append(dict(eos=eos, nu=nu, buffer_on=bool(buf), sink_model=model, rk_param=a))
    combos.append(dict(rk_param=0.75, rough=0.3))
    combos.append(dict(rk_param=1.0 / 3.0, moments=0.3))
    combos.append(dict(masses=False, eos=1, buffer_on=False, nu=0.0))
    combos.append(dict(masses=False, eos=1, buffer_on=True, nu=0.01, rk_param=0.5))
    combos.append(dict(sink_rate=-3.0, sink_model=1))
    combos.append(dict(ni=5, nj=3))
    combos.append(dict(ni=1, nj=1))
    combos.append(dict(ni=37, nj=131))
    return combos


def dg_limit_matrix():
    return [dict(), dict(moments=0.001), dict(moments=0.005), dict(moments=0.002, rough=0.002), dict(rough=0.3),
            dict(rough=0.3, moments=0.5), dict(big_slopes=True), dict(big_slopes=True, rough=0.5),
            dict(ni=1, nj=1, rough=0.5), dict(ni=37, nj=131, moments=0.003)]
