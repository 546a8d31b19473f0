"""
CPU checks against the committed golden fixtures (tests/golden/*.npz, produced by
running the reference itself -- see tests/golden/generate.py):

  * the host logic this package restates (setups' initial data, Kepler orbit,
    Physics.point_masses, mesh/patch extents, subdivide, the RK schedule, the
    stage-time formula, set_bc) is BIT-IDENTICAL to the reference's when driven
    with the CPU oracle kernels: initial data, every dt, every time and the final
    state must match the reference run exactly;
  * the oracle (oracle/sf_oracle.c) is thereby pinned a second time, end to end.
"""
import os

import numpy as np
import pytest

import ref_solver
from sailfish_amd.setups import make_setup

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    "iso_disk_64": ("circumbinary-disk", {}, {}),
    "iso_disk_128_p4": ("circumbinary-disk", {}, {}),
    "iso_disk_rk3_48": ("circumbinary-disk", {}, dict(rk_order=3)),
    "iso_explosion_64": ("cylindrical-explosion", {}, {}),
    "gam_disk_64": ("circumbinary-disk", dict(eos="gamma-law"), {}),
    "gam_cooling_48": ("circumbinary-disk", dict(eos="gamma-law", cooling_coefficient=1e-2), {}),
    "gam_explosion_64": ("cylindrical-explosion", dict(eos="gamma-law"), {}),
    # developed flows: the fixture's `initial` is the reference's state after 1500 steps, restarted as from a checkpoint
    "iso_disk_96_developed": ("circumbinary-disk", {}, {}),
    "gam_disk_96_developed": ("circumbinary-disk", dict(eos="gamma-law"), {}),
}


def start_of(g):
    """(time, solution) a solver is to be started with: the analytic initial data at t = 0 (solution None), or the
    developed state and time a fixture recorded."""
    if "time0" in g.files:
        return float(g["time0"]), g["initial"]
    return 0.0, None


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_solver_reproduces_reference_run_bitwise(name):
    setup_name, model, options = CASES[name]
    g = load(name)
    setup = make_setup(setup_name, **model)
    mesh = setup.mesh(int(g["resolution"]))
    time0, solution = start_of(g)
    s = ref_solver.OracleSolver(setup, mesh, time=time0, num_patches=int(g["num_patches"]), options=options,
                                solution=solution)
    assert np.array_equal(s.solution, g["initial"])
    cfl = float(g["cfl"])
    for dt_ref, t_ref in zip(g["dt"], g["time"]):
        dt = mesh.min_spacing() / s.maximum_wavespeed() * cfl
        assert dt == dt_ref
        s.advance(dt)
        assert s.time == t_ref
    assert np.array_equal(s.solution, g["final"])


def test_kepler_point_masses_bitwise():
    rows = load("kepler")["rows"]
    for e, q, t, x, y, vx, vy, m, rs, rate, rsink, model in rows:
        setup = make_setup("circumbinary-disk", eccentricity=float(e), mass_ratio=float(q))
        which = None
        for pm in setup.point_masses(float(t)):
            got = (pm.position_x, pm.position_y, pm.velocity_x, pm.velocity_y, pm.mass,
                   pm.softening_length, pm.sink_rate, pm.sink_radius, float(pm.sink_model.value))
            if got == (x, y, vx, vy, m, rs, rate, rsink, model):
                which = pm
        assert which is not None, (e, q, t)


def test_c_kepler_orbit_bitwise():
    """sfhip_kepler_orbital_state -- the orbit solved on the host in C for sfhip_plan_step_rk2, no GPU involved --
    gives the ten numbers of OrbitalElements.orbital_state bit for bit: on the reference-generated rows of
    kepler.npz and on a sweep of eccentric, unequal-mass orbits over many periods."""
    import ctypes
    import os

    from sailfish_amd.physics.kepler import OrbitalElements

    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sailfish_amd", "csrc",
                                   "libsfhip.so"))
    fn = lib.sfhip_kepler_orbital_state
    fn.argtypes = [ctypes.c_double] * 5 + [ctypes.POINTER(ctypes.c_double)]
    out = (ctypes.c_double * 10)()

    def c_state(a, m, q, e, t):
        assert fn(a, m, q, e, t, out) == 0
        return tuple(out)

    for e, q, t, x, y, vx, vy, m, *_ in load("kepler")["rows"]:
        got = c_state(1.0, 1.0, float(q), float(e), float(t))
        assert (x, y, vx, vy, m) in ((got[1], got[2], got[3], got[4], got[0]), (got[6], got[7], got[8], got[9], got[5]))
    rng = np.random.default_rng(4)
    for _ in range(3000):
        a, m, q, e = rng.uniform(0.5, 2.0), rng.uniform(0.5, 3.0), rng.uniform(0.05, 1.0), rng.uniform(0.0, 0.9)
        t = rng.uniform(-50.0, 500.0)
        b1, b2 = OrbitalElements(a, m, q, e).orbital_state(t)
        assert c_state(a, m, q, e, t) == (*b1, *b2)


@pytest.mark.parametrize("eos,n", [("isothermal", 96), ("gamma-law", 40), ("gamma-law", 600)])
def test_vectorised_initial_data_equals_per_point_loop(eos, n):
    setup = make_setup("circumbinary-disk", eos=eos)
    mesh = setup.mesh(n)
    rows = (3, 11) if n > 100 else (0, n)
    fast = setup.primitive_array(0.0, mesh, *rows)
    slow = np.zeros_like(fast)
    for i in range(*rows):
        for j in range(mesh.nj):
            setup.primitive(0.0, mesh.cell_coordinates(i, j), slow[i - rows[0], j])
    # (round 3: the gamma-law disk of any size comes from libsfhip.so's host routine, which makes the interpreter's own
    # libm calls per point -- no more "last place" allowance for large meshes)
    assert np.array_equal(fast, slow)


def test_gamma_law_initial_data_of_config_3_are_the_per_point_formulas():
    """BASELINE.json configs[3], cbdgam_2d 8192^2 (1024 x 8192 zones per rank): the initial data a rank builds
    (sfhip_host_gamma_law_disk_primitive: the per-point formulas with the interpreter's libm calls, threaded) against
    CircumbinaryDisk.primitive -- the reference's formulas, circumbinary_disk.py:94-108 -- at 4000 scattered zones of
    two ranks' slabs: identical bits.  numpy's vector pow / exp, which built these arrays until round 2, differ from the
    per-point values in the last place at about 5 % of the zones (also measured here, so that the reason for the native
    routine stays on record)."""
    setup = make_setup("circumbinary-disk", eos="gamma-law")
    mesh = setup.mesh(8192)
    rng = np.random.default_rng(7)
    point = np.zeros(4)
    vector_differs = 0
    for rows in ((0, 1024), (3072, 4096)):  # an edge slab and the one next to the binary
        slab = setup.primitive_array(0.0, mesh, *rows)
        assert slab.shape == (1024, 8192, 4)
        ii = rng.integers(rows[0], rows[1], 2000)
        jj = rng.integers(0, 8192, 2000)
        x = mesh.x0 + (ii + 0.5) * mesh.dx
        y = mesh.y0 + (jj + 0.5) * mesh.dy
        vector = setup._primitive_vectorised(x, y, x.shape)
        for k, (i, j) in enumerate(zip(ii, jj)):
            setup.primitive(0.0, mesh.cell_coordinates(int(i), int(j)), point)
            assert np.array_equal(slab[i - rows[0], j], point), (i, j)
            vector_differs += not np.array_equal(vector[k], point)
            assert np.allclose(vector[k], point, rtol=2e-15, atol=0)
    assert vector_differs > 0  # if numpy ever matches libm everywhere, the native routine is no longer needed


def _load_reductions(name):
    import json

    from sailfish_amd.physics.circumbinary import Diagnostic

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    diagnostics = []
    for d in json.loads(str(g["diagnostics"])):
        if d.get("radial_cut") is not None:
            d["radial_cut"] = tuple(d["radial_cut"])
        diagnostics.append(Diagnostic(**d))
    return g, diagnostics, json.loads(str(g["model"]))


@pytest.mark.parametrize("name", ["iso_reductions_64_p2", "gam_reductions_48"])
def test_oracle_reductions_reproduce_reference_bitwise(name):
    """tests/ref_solver.py's restatement of Solver.reductions() (numpy + oracle source-term kernels) against the
    values the reference itself produced for the same state (tests/golden/generate.py: run_reductions)."""
    g, diagnostics, model = _load_reductions(name)
    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(int(g["resolution"]))
    s = ref_solver.OracleSolver(setup, mesh, time=float(g["time"]), num_patches=int(g["num_patches"]),
                                solution=g["solution"])
    got = np.array(s.reductions(diagnostics), dtype=np.complex128)
    assert np.array_equal(got, g["values"])
