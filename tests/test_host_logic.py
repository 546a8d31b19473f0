"""
CPU tests of the host side: the C-ABI library loads and exports every declared
symbol, the header parser, the Library calling convention, subdivide helpers, and
the fused solver data flow (buffer rotation, clamps, u^n aliasing, wavespeed floor
and its fallback) driven by the oracle backend against the reference's runs.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from sailfish_amd.kernel.parse_api import parse_api
from sailfish_amd.subdivide import concat_on_host, lazy_reduce, partition, subdivide

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "sailfish_amd", "csrc", "libsfhip.so")
HEADER = os.path.join(ROOT, "include", "sfhip.h")


def test_header_parses_to_reference_signatures():
    api = parse_api(open(HEADER).read())
    # argument counts of the reference's kernels (probed with the reference's parse_api)
    assert len(api["cbdiso_2d_advance_rk"].args) == 41
    assert len(api["cbdiso_2d_primitive_to_conserved"].args) == 4
    assert len(api["cbdiso_2d_wavespeed"].args) == 29
    assert len(api["cbdiso_2d_point_mass_source_term"].args) == 17
    assert len(api["cbdgam_2d_advance_rk"].args) == 44
    assert len(api["cbdgam_2d_wavespeed"].args) == 5
    assert len(api["cbdgam_2d_primitive_to_conserved"].args) == 5
    assert len(api["cbdgam_2d_point_mass_source_term"].args) == 29
    assert all(s.rank == 2 for n, s in api.items() if n.startswith("cbd"))
    a = api["cbdiso_2d_advance_rk"].args
    assert [x.name for x in a[:9]] == ["ni", "nj", "patch_xl", "patch_xr", "patch_yl", "patch_yr",
                                       "conserved_rk", "primitive_rd", "primitive_wr"]
    assert a[6].dtype == "double*" and a[6].constraint == "$.shape == (ni + 4, nj + 4, 3)"
    assert [x.name for x in a[-5:]] == ["nu", "a", "dt", "velocity_ceiling", "density_floor"]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference sources not present")
def test_header_matches_reference_kernel_signatures_exactly():
    """Every reference-named declaration has the reference's argument names and types, in order."""
    import ast

    api = parse_api(open(HEADER).read())
    ours = {n: [(a.dtype, a.name) for a in s.args] for n, s in api.items() if not n.startswith("sfhip_")}
    seen = set()
    pat = __import__("re").compile(r"\s*(\w+\s*\**)\s*(\w+)\s*[,\)]")
    for solver in ("cbdiso_2d", "cbdgam_2d", "srhd_2d", "cbdisodg_2d"):
        lines = open(f"/root/reference/sailfish/solvers/{solver}.c").read().splitlines()
        k = 0
        while k < len(lines):
            if lines[k].startswith("PUBLIC void"):
                name = lines[k].split()[2].rstrip("(")
                args = []
                k += 1
                while True:
                    m = pat.match(lines[k])
                    args.append((m.group(1).replace(" ", ""), m.group(2)))
                    if ")" in lines[k].split("//")[0]:
                        break
                    k += 1
                assert ours[name] == args, name
                seen.add(name)
            k += 1
    assert seen == set(ours)  # every PUBLIC kernel of the four solvers is declared, and nothing else borrows their names
    del ast


def test_library_exports_every_declared_symbol():
    api = parse_api(open(HEADER).read())
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    for name in list(api) + ["sfhip_set_device", "sfhip_device_count", "sfhip_set_stream", "sfhip_synchronize",
                             "sfhip_last_error", "sfhip_selftest_lane_shift",
                             "sfhip_plan_create_cbdiso_2d", "sfhip_plan_create_cbdgam_2d", "sfhip_plan_set_side_stream",
                             "sfhip_plan_destroy", "sfhip_plan_stage", "sfhip_plan_host_wait", "sfhip_plan_step_rk2", "sfhip_plan_step_timing",
                             "sfhip_kepler_orbital_state"]:
        assert name in exported, name


def test_library_loads_and_refuses_bad_calls_without_gpu():
    from sailfish_amd.kernel.library import KernelError, Library

    lib = Library(mode="hip", debug=True)
    assert lib.module.sfhip_device_count() >= 0
    with pytest.raises(ValueError):
        lib.cbdiso_2d_wavespeed[(4, 4, 4)]
    with pytest.raises(AttributeError):
        lib.no_such_kernel
    with pytest.raises(ValueError):
        Library(mode="cpu")
    with pytest.raises(KernelError):
        Library(mode="hip", path="/nonexistent/libsfhip.so")


def test_solver_refuses_modes_without_kernels():
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import SolverInitializationError, make_solver

    setup = make_setup("circumbinary-disk")
    with pytest.raises(SolverInitializationError):
        make_solver("cbdiso_2d", setup.physics, {}, setup=setup, mesh=setup.mesh(16), mode="cpu")
    with pytest.raises(SolverInitializationError):
        make_solver("srhd_1d", {}, {})


def test_subdivide_helpers():
    assert partition(10, 3) == [4, 3, 3]
    assert subdivide(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert subdivide((5, 11), 2) == [(5, 8), (8, 11)]
    assert subdivide(4096, 8)[-1] == (3584, 4096)
    import contextlib

    calls = []
    r = lazy_reduce(max, float, (lambda k=k: calls.append(k) or k * 1.5 for k in range(4)),
                    (contextlib.nullcontext() for _ in range(4)))
    assert r == 4.5 and calls == [0, 1, 2, 3]
    a = np.arange(6 * 7 * 2, dtype=float).reshape(6, 7, 2)
    b = a + 100
    c = concat_on_host([a, b], (2, 2))
    assert c.shape == (4, 3, 2)
    assert np.array_equal(c[:2], a[2:-2, 2:-2]) and np.array_equal(c[2:], b[2:-2, 2:-2])
    assert concat_on_host([a[..., 0], b[..., 0]], 1, rank=1).shape == (8, 7)


@pytest.mark.parametrize("name,num_patches", [
    ("iso_disk_64", 1), ("iso_disk_128_p4", 4), ("iso_disk_rk3_48", 3), ("iso_explosion_64", 2),
    ("gam_disk_64", 1), ("gam_cooling_48", 1), ("gam_explosion_64", 4),
])
def test_fused_data_flow_reproduces_reference_bitwise(name, num_patches):
    """Fused solver + oracle kernels == the reference's run: buffer rotation, in-place last stage,
    clamped boundaries, single-process halos and the fused wavespeed (with floor) are all exact."""
    from oracle_backend import OracleBackend
    from sailfish_amd.driver import timestep
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver
    from test_golden_cpu import CASES, load

    setup_name, model, options = CASES[name]
    g = load(name)
    setup = make_setup(setup_name, **model)
    mesh = setup.mesh(int(g["resolution"]))
    s = make_solver(setup.solver, setup.physics, options, setup=setup, mesh=mesh, time=0.0,
                    num_patches=num_patches, backend=OracleBackend(), distributed=False)
    for dt_ref, t_ref in zip(g["dt"], g["time"]):
        dt = timestep(mesh, s, float(g["cfl"]))
        assert dt == dt_ref
        s.advance(dt)
        assert s.time == t_ref
    assert np.array_equal(s.solution, g["final"])


def test_wavespeed_floor_fallback():
    """If the fused maximum comes out below the floor the solver must recompute it exactly."""
    from oracle_backend import OracleBackend
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver

    setup = make_setup("circumbinary-disk")
    mesh = setup.mesh(32)
    s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, backend=OracleBackend(),
                    distributed=False)
    a0 = s.maximum_wavespeed()
    s._wavespeed_floor = 100.0 * a0  # nothing can reach this
    s.advance(1e-3)
    assert float(s.patches[0].wavespeed_max[0]) == 0.0  # fused epilogue folded nothing
    a1 = s.maximum_wavespeed()
    fresh = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=s.time, solution=s.solution,
                        backend=OracleBackend(), distributed=False)
    assert a1 == fresh.maximum_wavespeed() and a1 > 0.0


@pytest.mark.parametrize("name,num_patches", [
    ("iso_disk_64", 1), ("iso_disk_128_p4", 4), ("iso_disk_rk3_48", 3), ("gam_disk_64", 1), ("gam_explosion_64", 4),
])
def test_step_with_device_side_dt_reproduces_reference_bitwise(name, num_patches):
    """Solver.step(cfl) -- dt formed on the device, read-back overlapped with stage 0 -- gives the same
    dt sequence, times and final state as the reference's `dt = dx / amax * cfl; advance(dt)`."""
    from oracle_backend import OracleBackend
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver
    from test_golden_cpu import CASES, load

    setup_name, model, options = CASES[name]
    g = load(name)
    setup = make_setup(setup_name, **model)
    mesh = setup.mesh(int(g["resolution"]))
    s = make_solver(setup.solver, setup.physics, options, setup=setup, mesh=mesh, time=0.0,
                    num_patches=num_patches, backend=OracleBackend(), distributed=False)
    for dt_ref, t_ref in zip(g["dt"], g["time"]):
        assert s.step(float(g["cfl"])) == dt_ref
        assert s.time == t_ref
    assert np.array_equal(s.solution, g["final"])


def test_step_redoes_stage_when_filtered_wavespeed_is_below_floor():
    """step() launches stage 0 before it knows the wavespeed; a filtered (too small) value must be
    recomputed exactly and the stage redone, leaving the same state as the blocking path."""
    from oracle_backend import OracleBackend
    from sailfish_amd.driver import timestep
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver

    setup = make_setup("circumbinary-disk")
    mesh = setup.mesh(32)
    kw = dict(setup=setup, mesh=mesh, time=0.0, backend=OracleBackend(), distributed=False)
    a = make_solver(setup.solver, setup.physics, {}, **kw)
    b = make_solver(setup.solver, setup.physics, {}, **kw)
    for s in (a, b):
        s.maximum_wavespeed()
        s._wavespeed_floor = 1e6  # the fused epilogue of the next advance folds nothing
        s.advance(1e-3)
        assert float(s.patches[0].wavespeed_max[0]) == 0.0
    dt_a = a.step(0.3)
    dt_b = timestep(mesh, b, 0.3)
    b.advance(dt_b)
    assert dt_a == dt_b and a.time == b.time
    assert np.array_equal(a.solution, b.solution)


@pytest.mark.parametrize("name", ["iso_reductions_64_p2", "gam_reductions_48"])
def test_solver_reductions_reproduce_reference_bitwise(name):
    """Solver.reductions() host logic (spec encoding, batching by 8, patch order, dx*dy, complex results) with
    the oracle backend == the values the reference produced for the same state."""
    from oracle_backend import OracleBackend
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver
    from test_golden_cpu import _load_reductions

    g, diagnostics, model = _load_reductions(name)
    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(int(g["resolution"]))
    physics = dict(setup.physics)
    physics["diagnostics"] = diagnostics
    s = make_solver(setup.solver, physics, {}, setup=setup, mesh=mesh, time=float(g["time"]),
                    solution=g["solution"], num_patches=int(g["num_patches"]), backend=OracleBackend(),
                    distributed=False)
    got = s.reductions()
    assert [type(v) for v in got] == [complex if d.quantity in ("sigma_m1", "eccentricity_vector") else float
                                      for d in diagnostics]
    assert np.array_equal(np.array(got, dtype=np.complex128), g["values"])


def test_reductions_reject_what_the_reference_cannot_evaluate():
    from oracle_backend import OracleBackend
    from sailfish_amd.physics.circumbinary import Diagnostic
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver

    setup = make_setup("circumbinary-disk")
    mesh = setup.mesh(16)
    for bad in (Diagnostic("mdot", which_mass=1), Diagnostic("power", gravity=True, which_mass="both"),
                Diagnostic("nonsense"), Diagnostic(3, gravity=True, which_mass=1),
                Diagnostic("torque", gravity=True, which_mass=3)):
        physics = dict(setup.physics, diagnostics=[bad])
        s = make_solver(setup.solver, physics, {}, setup=setup, mesh=mesh, time=0.0, backend=OracleBackend(),
                        distributed=False)
        with pytest.raises(ValueError):
            s.reductions()


MATCHING_CODE = """
#define NCONS 3
static double helper(double x) { return x; }

PUBLIC void cbdiso_2d_primitive_to_conserved(
    int ni,
    int nj,
    double *primitive, // :: $.shape == (ni + 4, nj + 4, 3)
    double *conserved) // :: $.shape == (ni + 4, nj + 4, 3)
{
    int ng = 2;
}

PUBLIC void cbdgam_2d_wavespeed(int ni, int nj,
    double *primitive, double *wavespeed,
    double gamma_law_index)
{
}
"""


def test_library_checks_the_code_it_is_handed():
    """``Library(code, mode=...)`` is how a reference solver makes its kernel library (library.py:165-185,
    cbdiso_2d.py:303-305).  Nothing is compiled in the hip mode, but the ``PUBLIC void`` declarations of the code are
    checked against the library: same kernels, same argument types and names, same order -- a binding learns at
    construction, not from a wrong result, that the text it hands over is not what the library implements."""
    from sailfish_amd.kernel.library import KernelError, Library
    from sailfish_amd.kernel.parse_api import parse_reference_api

    api = parse_reference_api(MATCHING_CODE)
    assert sorted(api) == ["cbdgam_2d_wavespeed", "cbdiso_2d_primitive_to_conserved"]
    assert [a.name for a in api["cbdgam_2d_wavespeed"].args] == ["ni", "nj", "primitive", "wavespeed", "gamma_law_index"]
    assert api["cbdiso_2d_primitive_to_conserved"].args[2].constraint == "$.shape == (ni + 4, nj + 4, 3)"
    assert api["cbdiso_2d_primitive_to_conserved"].rank == 2
    lib = Library(MATCHING_CODE, mode="hip", debug=False)
    assert lib.declared == sorted(api)
    for wrong, what in (
        (MATCHING_CODE.replace("double *conserved)", "double *conserved,\n    double extra)"), "arguments"),
        (MATCHING_CODE.replace("double gamma_law_index", "int gamma_law_index"), "argument 4"),
        (MATCHING_CODE.replace("double *primitive, double *wavespeed", "double *wavespeed, double *primitive"), "argument 2"),
        (MATCHING_CODE.replace("cbdgam_2d_wavespeed", "cbdgam_2d_no_such_kernel"), "does not provide"),
        (MATCHING_CODE.replace("int nj,\n    double *primitive, //", "int nj,\n    float *primitive, //"), "cannot read"),
    ):
        with pytest.raises(KernelError, match=what):
            Library(wrong, mode="hip")
    # macros: the srhd_2d kernels' two are settings of the prebuilt library; anything else cannot be honoured
    Library(mode="hip", define_macros=dict(PLM_THETA=2.0, MACH_CEILING=100.0))
    Library(mode="hip", define_macros=dict(PLM_THETA=1.5, MACH_CEILING=1e6))  # (back to the reference's defaults)
    with pytest.raises(KernelError, match="prebuilt"):
        Library(mode="hip", define_macros=dict(NCONS=4))
    with pytest.raises(KernelError):
        Library(mode="hip", define_macros=dict(PLM_THETA=3.0))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference sources not present")
@pytest.mark.parametrize("solver", ["cbdiso_2d", "cbdgam_2d", "srhd_2d", "cbdisodg_2d"])
def test_library_accepts_the_reference_solvers_own_code(solver):
    """The text the reference's four solvers pass to Library -- their .c files, read where they lie -- passes the check."""
    from sailfish_amd.kernel.library import Library

    with open(f"/root/reference/sailfish/solvers/{solver}.c") as f:
        lib = Library(f.read(), mode="hip", debug=False)
    assert lib.declared and all(name.startswith(solver) for name in lib.declared)
