"""
Solver-level GPU parity: the device-resident Solver (fused and reference-order
paths, one and several patches, RK1/2/3) against the reference's solver loop
run on the CPU oracle -- same initial data, N steps, dt sequence bit-exact,
solution within 1e-12 relative per zone.
"""
import numpy as np
import pytest

import ref_solver
from sailfish_amd.driver import timestep
from sailfish_amd.setups import make_setup
from sailfish_amd.solvers import make_solver

pytestmark = pytest.mark.gpu
TOL = 1e-12


# Each field against its own magnitude (cases.component_relative_error).  A component below 1e-6 of its zone's
# scale is measured against that floor, so this bar is implied by the per-zone one (1e-12 / 1e-6); the figure is
# what matters and is appended to gpurun_out/parity_figures.txt for DESIGN.md.
COMPONENT_TOL = 1e-6


def zone_err(a, b):
    return float((np.abs(a - b).max(axis=-1) / np.maximum(np.abs(b).max(axis=-1), 1e-300)).max())


def check(got, expect, tol=TOL):
    """north_star's bar (1e-12 relative per zone) plus the per-component figure, printed with `pytest -s`."""
    import cases

    zerr = zone_err(got, expect)
    cerr = cases.component_relative_error(got, expect, interior=False)
    line = f"max relative error per zone {zerr:.2e}, per component {cerr:.2e}"
    print(line)
    try:
        import os

        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_figures.txt"), "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?')}: {line}\n")
    except OSError:
        pass
    assert zerr <= tol, zerr
    assert cerr <= COMPONENT_TOL, cerr


def run_pair(setup_name, n, steps, model=None, options=None, num_patches=1, fused=True, cfl=None):
    """
    Free-running pair: each solver computes its own dt from its own state.

    dt is asserted BIT-EXACT (a) at step 0, where both hold the same initial data, and (b) at every
    step for a HIP solver restarted from the CPU trajectory's state (`solution=`), i.e. the
    wavespeed / max / dt path is bit-exact for identical input along the whole N-step trajectory.
    The free-running GPU state differs from the CPU one in the last place (libm pow/exp vs the
    device's), so its own dt is compared to 1e-13 relative instead.
    """
    setup = make_setup(setup_name, **(model or {}))
    mesh = setup.mesh(n)
    kw = dict(setup=setup, mesh=mesh, num_patches=num_patches, mode="hip", fused=fused)
    gpu = make_solver(setup.solver, setup.physics, options or {}, time=0.0, **kw)
    cpu = ref_solver.OracleSolver(setup, mesh, time=0.0, num_patches=num_patches, options=options)
    cfl = cfl or gpu.recommended_cfl
    for step in range(steps):
        dt_gpu = timestep(mesh, gpu, cfl)
        dt_cpu = mesh.min_spacing() / cpu.maximum_wavespeed() * cfl
        if step == 0:
            assert dt_gpu == dt_cpu
        else:
            assert abs(dt_gpu - dt_cpu) <= 1e-13 * dt_cpu
            forced = make_solver(setup.solver, setup.physics, options or {}, time=cpu.time,
                                 solution=cpu.solution, **kw)
            assert timestep(mesh, forced, cfl) == dt_cpu  # bit-exact dt on identical state
        gpu.advance(dt_gpu)
        cpu.advance(dt_cpu)
        assert abs(gpu.time - cpu.time) <= 1e-13 * abs(cpu.time)
    return gpu.solution, cpu.solution


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("num_patches", [1, 4])
def test_iso_circumbinary_disk(fused, num_patches):
    got, expect = run_pair("circumbinary-disk", 128, 20, num_patches=num_patches, fused=fused)
    assert got.shape == (128, 128, 3)
    check(got, expect)


@pytest.mark.parametrize("rk_order", [1, 3])
def test_iso_rk_orders(rk_order):
    got, expect = run_pair("circumbinary-disk", 96, 6, options=dict(rk_order=rk_order), num_patches=3)
    check(got, expect)


def test_iso_multi_patch_bitwise_equals_single_patch():
    """Power-of-two meshes are decomposition invariant in the reference."""
    one, _ = run_pair("circumbinary-disk", 128, 5, num_patches=1)
    four, _ = run_pair("circumbinary-disk", 128, 5, num_patches=4)
    assert np.array_equal(one, four)


def test_iso_cylindrical_explosion():
    got, expect = run_pair("cylindrical-explosion", 64, 20)
    check(got, expect)


def test_iso_eccentric_unequal_binary():
    got, expect = run_pair("circumbinary-disk", 96, 10, model=dict(eccentricity=0.3, mass_ratio=0.2, nu=0.0,
                                                                 sink_model="force_free"))
    check(got, expect)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("num_patches", [1, 3])
def test_gam_circumbinary_disk(fused, num_patches):
    got, expect = run_pair("circumbinary-disk", 96, 20, model=dict(eos="gamma-law"), num_patches=num_patches,
                           fused=fused)
    assert got.shape == (96, 96, 4)
    check(got, expect)


def test_gam_cooling_and_explosion():
    got, expect = run_pair("circumbinary-disk", 64, 10, model=dict(eos="gamma-law", cooling_coefficient=1e-2))
    check(got, expect)
    got, expect = run_pair("cylindrical-explosion", 64, 20, model=dict(eos="gamma-law"))
    check(got, expect)


def test_restart_from_solution_is_bitwise():
    """``solution=`` round trip (checkpoint layout contract, driver.py:367,446): (ni, nj, 3) host array."""
    setup = make_setup("circumbinary-disk")
    mesh = setup.mesh(64)
    a = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
    dts = []
    for _ in range(3):
        dts.append(timestep(mesh, a, 0.3))
        a.advance(dts[-1])
    chk, t = a.solution, a.time
    b = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=t, solution=chk, mode="hip")
    for s in (a, b):
        dt = timestep(mesh, s, 0.3)
        s.advance(dt)
    assert np.array_equal(a.solution, b.solution)


@pytest.mark.parametrize("name,model,num_patches", [
    ("circumbinary-disk", {}, 1), ("circumbinary-disk", {}, 4), ("circumbinary-disk", dict(eos="gamma-law"), 2),
])
def test_step_equals_timestep_then_advance_bitwise(name, model, num_patches):
    """Solver.step(cfl) (dt formed on the device, read-back overlapped with stage 0) == the reference
    driver's two blocking calls (driver.py:508-511), bit for bit, dt sequence included."""
    setup = make_setup(name, **model)
    mesh = setup.mesh(96)
    kw = dict(setup=setup, mesh=mesh, time=0.0, num_patches=num_patches, mode="hip")
    a = make_solver(setup.solver, setup.physics, {}, **kw)
    b = make_solver(setup.solver, setup.physics, {}, **kw)
    cfl = a.recommended_cfl
    for _ in range(8):
        dt_a = a.step(cfl)
        dt_b = timestep(mesh, b, cfl)
        b.advance(dt_b)
        assert dt_a == dt_b and a.time == b.time
    assert np.array_equal(a.solution, b.solution)


def test_step_redoes_stage_when_filtered_wavespeed_is_below_floor():
    setup = make_setup("circumbinary-disk")
    mesh = setup.mesh(64)
    kw = dict(setup=setup, mesh=mesh, time=0.0, mode="hip")
    a = make_solver(setup.solver, setup.physics, {}, **kw)
    b = make_solver(setup.solver, setup.physics, {}, **kw)
    for s in (a, b):
        s.maximum_wavespeed()
        s._wavespeed_floor = 1e6  # the fused epilogue of the next advance folds nothing
        s.advance(1e-3)
    dt_a = a.step(0.3)
    dt_b = timestep(mesh, b, 0.3)
    b.advance(dt_b)
    assert dt_a == dt_b and a.time == b.time
    assert np.array_equal(a.solution, b.solution)


@pytest.mark.parametrize("eos", ["isothermal", "gamma-law"])
def test_full_physics_instantiation(eos, monkeypatch):
    """The full-physics instantiations of the stage kernels (flag tests resolved on the host)
    evaluate the generic instantiations' formulas; the compiler contracts a few of them into FMAs
    differently, so the trajectories agree to the last places, not bit for bit (SFHIP_GENERIC_STAGE=1
    forces the generic ones; the variable is read at every launch).  All patches and ranks of a run use
    the same instantiation, so decomposition invariance (test_hip_reductions) is not affected."""
    setup = make_setup("circumbinary-disk", eos=eos)
    mesh = setup.mesh(192)
    runs = []
    for generic in ("0", "1"):
        monkeypatch.setenv("SFHIP_GENERIC_STAGE", generic)
        s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, num_patches=3, mode="hip")
        dts = [s.step(s.recommended_cfl) for _ in range(6)]
        runs.append((dts, s.solution))
    assert runs[0][0][0] == runs[1][0][0]  # same initial state: bit-exact dt
    assert max(abs(a - b) / b for a, b in zip(*[r[0] for r in runs])) <= 1e-13
    assert zone_err(runs[0][1], runs[1][1]) <= 1e-13


def test_gam_decomposition_invariance_2048():
    """cbdgam_2d at a size with many row chunks per patch (chunk boundaries fall on different rows in the
    1-patch and 8-patch runs; 8 x 256 rows is the per-GPU shape of BASELINE configs[3] scaled down): slab
    decomposition changes neither dt nor a single bit of the state."""
    setup = make_setup("circumbinary-disk", eos="gamma-law")
    mesh = setup.mesh(2048)
    runs = []
    for num_patches in (1, 8):
        s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, num_patches=num_patches,
                        mode="hip")
        dts = [s.step(s.recommended_cfl) for _ in range(3)]
        runs.append((dts, s.solution))
    assert runs[0][0] == runs[1][0]
    assert np.array_equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("model,num_patches,rk_order", [({}, 1, 2), ({}, 3, 3), (dict(eos="gamma-law"), 2, 2)])
def test_launch_plans_equal_library_calls(model, num_patches, rk_order, monkeypatch):
    """The solvers launch the fused stage through a plan (constants marshalled once, include/sfhip.h "launch
    plans"); SFHIP_NO_PLANS=1 sends the same launches through Library's kernel[shape](*args).  Same C function
    underneath: dt sequence and state agree bit for bit, for step() and for timestep() + advance()."""
    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(96)
    options = dict(rk_order=rk_order) if rk_order != 2 else {}
    runs = []
    for no_plans in ("", "1"):
        monkeypatch.setenv("SFHIP_NO_PLANS", no_plans)
        s = make_solver(setup.solver, setup.physics, options, setup=setup, mesh=mesh, time=0.0,
                        num_patches=num_patches, mode="hip")
        assert (s.patches[0].plan is None) == bool(no_plans)
        dts = [s.step(s.recommended_cfl) for _ in range(4)]
        dts.append(timestep(mesh, s, s.recommended_cfl))
        s.advance(dts[-1])
        dts += [s.step(s.recommended_cfl) for _ in range(2)]
        runs.append((dts, s.time, s.solution))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]
    assert np.array_equal(runs[0][2], runs[1][2])


@pytest.mark.parametrize("model", [{}, dict(eos="gamma-law"), dict(eccentricity=0.4, mass_ratio=0.3)],
                         ids=["iso", "gam", "eccentric"])
def test_one_call_step_equals_the_python_step(model, monkeypatch):
    """sfhip_plan_step_rk2 (both stages, dt, stage times and the Kepler orbit from one C call) against the same
    step driven from Python (SFHIP_NO_C_STEP=1): dt sequence, times and state bit for bit -- also across the
    redo that a too-small filtered wavespeed forces, and when the two kinds of step alternate."""
    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(96)
    runs = []
    for no_c_step in ("", "1"):
        monkeypatch.setenv("SFHIP_NO_C_STEP", no_c_step)
        s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
        assert s._c_step == (not no_c_step)
        dts = [s.step(s.recommended_cfl) for _ in range(5)]
        s._wavespeed_floor = 1e6               # the next step's wavespeed looks filtered away: exact recompute + redo
        dts += [s.step(s.recommended_cfl) for _ in range(2)]
        dts.append(timestep(mesh, s, s.recommended_cfl))  # the blocking API in between
        s.advance(dts[-1])
        dts += [s.step(s.recommended_cfl) for _ in range(2)]
        runs.append((dts, s.time, s.solution))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]
    assert np.array_equal(runs[0][2], runs[1][2])


@pytest.mark.parametrize("model", [{}, dict(eos="gamma-law")], ids=["iso", "gam"])
def test_wavespeed_filter_fraction_does_not_change_the_bits(model, monkeypatch):
    """The fused epilogue evaluates exactly only the zones that reach a fraction of the previous maximum, and a step
    whose maximum falls below that bar is repeated from an exact recomputation: whatever the fraction -- none at
    all, the default, one so high that steps do repeat -- dt sequence and state are the same, bit for bit."""
    from sailfish_amd.solvers import cbd_common

    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(96)
    runs, redos = [], []
    for fraction in (0.0, cbd_common.WAVESPEED_FLOOR_FRACTION, 0.9999):
        monkeypatch.setattr(cbd_common, "WAVESPEED_FLOOR_FRACTION", fraction)
        s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
        dts = [s.step(s.recommended_cfl) for _ in range(12)]
        runs.append((dts, s.time, s.solution))
        redos.append(s.wavespeed_redos)
    assert redos[0] == 0 and redos[2] > 0, redos  # the highest bar is missed whenever the maximum falls at all
    for other in runs[1:]:
        assert other[0] == runs[0][0] and other[1] == runs[0][1] and np.array_equal(other[2], runs[0][2])


@pytest.mark.parametrize("model", [{}, dict(eos="gamma-law")], ids=["iso", "gam"])
@pytest.mark.parametrize("no_c_step", ["", "1"], ids=["one-call", "python-stages"])
def test_blown_up_state_kills_dt_not_the_process(no_c_step, model, monkeypatch):
    """A NaN in the state must end in a dt that visibly dies, as in the reference, not in a finite-looking one or an
    exception of the step's own making.  cbdgam: the NaN energy stays a NaN pressure (`max2(floor, p)`), reaches the
    wavespeed fold, the next dt is NaN and the step fails where the reference's does.  cbdiso: the reference's recovery turns the NaN zone into floor density
    at ceiling velocity (cbdiso_2d.c:268-283), so the wavespeed jumps to the velocity ceiling and dt collapses."""
    import math

    monkeypatch.setenv("SFHIP_NO_C_STEP", no_c_step)
    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(64)
    s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
    healthy = s.step(s.recommended_cfl)
    assert math.isfinite(healthy)
    s.patches[0].primitive1[20, 30, 0] = float("nan")
    assert math.isfinite(s.step(s.recommended_cfl))  # formed from the wavespeed of the state before the NaN appeared
    if model:
        # dt is NaN now; the step ends as the reference's advance(dt) does: floor(NaN) in the Kepler solve of stage 1's time
        with pytest.raises(ValueError, match="NaN"):
            s.step(s.recommended_cfl)
    else:
        assert s.step(s.recommended_cfl) < 1e-6 * healthy and s.step(s.recommended_cfl) < 1e-6 * healthy


@pytest.mark.parametrize("model,spinup", [({}, 4000), (dict(eos="gamma-law"), 6000)], ids=["iso", "gam"])
def test_pair_from_a_developed_flow_at_384(model, spinup):
    """The oracle and the HIP solver side by side from a DEVELOPED state at a size the analytic-start pairs do not
    reach: the device runs some thousand steps at 384^2 (streams into the sinks, shocks, the buffer driving the
    edge), both solvers are then started from that state and time, as from a checkpoint, and take 8 steps each with
    its own dt: first dt bit-identical, state within 1e-12 per zone."""
    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(384)
    first = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
    cfl = first.recommended_cfl
    for _ in range(spinup):
        first.step(cfl)
    state, t0 = first.solution, first.time
    assert np.isfinite(state).all() and t0 > 3.0
    del first
    gpu = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=t0, solution=state, mode="hip",
                      num_patches=3)
    cpu = ref_solver.OracleSolver(setup, mesh, time=t0, solution=state, num_patches=3)
    for step in range(8):
        dt_cpu = mesh.min_spacing() / cpu.maximum_wavespeed() * cfl
        dt_gpu = gpu.step(cfl)
        if step == 0:
            assert dt_gpu == dt_cpu
        else:
            assert abs(dt_gpu - dt_cpu) <= 1e-13 * dt_cpu
        cpu.advance(dt_cpu)
        assert abs(gpu.time - cpu.time) <= 1e-13 * abs(cpu.time)
    check(gpu.solution, cpu.solution)


@pytest.mark.parametrize("model", [dict(domain_radius=3.3), dict(domain_radius=3.3, eos="gamma-law")], ids=["iso", "gam"])
def test_wavespeed_is_bit_exact_on_a_mesh_whose_spacing_is_not_dyadic(model):
    """Every other mesh in these tests spans [-12, 12] or a power of two, where (i + 0.5) dx is exact and a fused
    multiply-add cannot differ from the reference's two roundings.  Here dx = 6.6 / 100: the zone coordinates the
    fused epilogue's exact path uses must be the reference's `xl + (i + 0.5) * dx` in two roundings (`mesh_coord`,
    round 3), or the reduced wavespeed differs from `cbdiso_2d_wavespeed` + max in the last place.  The fused
    epilogue (state after a step), the standalone kernel on the same state and the oracle's kernel on the host copy
    of that state give the same bits; so does dt."""
    import oracle

    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(100)
    s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip", num_patches=3)
    cfl = s.recommended_cfl
    for _ in range(5):
        s.step(cfl)
    fused = s.maximum_wavespeed()                      # folded by the last stage's epilogue
    standalone = s._reduce_wavespeed(recompute=True)   # the reference-named wavespeed kernel + max
    assert fused == standalone
    cpu = ref_solver.OracleSolver(setup, mesh, time=s.time, solution=s.solution, num_patches=3)
    assert cpu.maximum_wavespeed() == fused
    assert timestep(mesh, s, cfl) == mesh.min_spacing() / fused * cfl
