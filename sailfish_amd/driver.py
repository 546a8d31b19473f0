"""
The part of the reference driver that is on the hot path: the fold loop with the
CFL timestep and the Mzps metric (``sailfish/driver.py:506-517``).

Everything else in the reference's driver (CLI, events, pickle checkpoints,
logging, config files) is control plane and out of scope;
the reference's own driver can drive these solvers through its solver-extension
hook.
"""
from typing import NamedTuple

from .kernel.system import measure_time
from .setups import make_setup
from .solvers import make_solver


class FoldReport(NamedTuple):
    iteration: int
    time: float
    dt: float
    mzps: float
    seconds: float


def timestep(mesh, solver, cfl_number, siml_time=None):
    """``dt = dx / amax * cfl`` in this operation order (driver.py:509-510): dt must match bit-for-bit."""
    dx = mesh.min_spacing(siml_time)
    return dx / solver.maximum_wavespeed() * cfl_number


def simulate(solver, mesh, cfl_number=None, fold=10, new_timestep_cadence=1, max_iterations=None,
             end_time=None, reference_time=1.0, iteration=0, dt=None):
    """
    Generator over folds: advances `fold` iterations, then yields a FoldReport.
    Mzps = zones * fold / fold_seconds * 1e-6 (driver.py:514), the clock read
    after every device holding a patch has drained.
    """
    cfl_number = cfl_number or solver.recommended_cfl
    if cfl_number > solver.maximum_cfl:
        raise ValueError(f"cfl number {cfl_number} is greater than {solver.maximum_cfl}")
    mode = getattr(solver, "mode", "host")
    while True:
        siml_time = solver.time
        if end_time is not None and siml_time / reference_time >= end_time:
            return
        if max_iterations is not None and iteration >= max_iterations:
            return
        steps = fold if max_iterations is None else min(fold, max_iterations - iteration)
        with measure_time(mode) as fold_time:
            for _ in range(steps):
                if new_timestep_cadence == 1 and hasattr(solver, "step"):
                    dt = solver.step(cfl_number, mesh.min_spacing(siml_time))  # same dt, read-back overlapped
                else:
                    if dt is None or (iteration % new_timestep_cadence == 0):
                        dt = timestep(mesh, solver, cfl_number, siml_time)
                    solver.advance(dt)
                iteration += 1
        seconds = fold_time()
        # srhd_2d: where the reference's C path prints [FATAL] and exits from inside the kernel
        # (srhd_2d.c:164-184) the device counts the zone; surface that once per fold, outside the timed
        # region (one 4-byte read-back), instead of carrying on with garbage primitives
        if hasattr(solver, "check_failures"):
            solver.check_failures()
        mzps = mesh.num_total_zones / seconds * 1e-6 * steps
        yield FoldReport(iteration, solver.time, dt, mzps, seconds)


def run(setup_name, resolution=None, num_patches=None, mode="hip", model_parameters=None, solver_options=None,
        cfl_number=None, fold=10, new_timestep_cadence=1, max_iterations=None, end_time=None, timeseries=None,
        **solver_kwargs):
    """
    Build setup, mesh and solver, run to completion, return (solver, reports).
    Counterpart of ``sailfish.driver.run`` (driver.py:522-543) without events,
    except the one that is arithmetic on the solution: when `timeseries` is a
    list, ``solver.reductions()`` is appended to it after every fold (the
    reference's ``timeseries`` event, driver.py:498-499, at fold cadence).
    """
    setup = make_setup(setup_name, **(model_parameters or {}))
    mesh = setup.mesh(resolution or setup.default_resolution)
    solver = make_solver(
        setup.solver, setup.physics, solver_options or {},
        setup=setup, mesh=mesh, time=setup.start_time, solution=None,
        num_patches=num_patches, mode=mode, **solver_kwargs,
    )
    if end_time is None and max_iterations is None:
        end_time = setup.default_end_time
    reports = []
    for report in simulate(solver, mesh, cfl_number, fold, new_timestep_cadence, max_iterations, end_time,
                           setup.reference_time_scale):
        reports.append(report)
        if timeseries is not None:
            timeseries.append(solver.reductions())
    return solver, reports
