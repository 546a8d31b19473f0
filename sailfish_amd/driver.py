"""
The part of the reference driver that is on the hot path -- the fold loop with the
CFL timestep and the Mzps metric (``sailfish/driver.py:506-517``) -- and the data
format on its far side: the checkpoint dict (``driver.py:143-189``) with restart
from it (``driver.py:343-384, 439-449``), SURVEY.md section 8(f) rank 2.

Everything else in the reference's driver (CLI, events, logging, config files) is
control plane and out of scope; the reference's own driver
can drive these solvers through its solver-extension hook.
"""
import os
import pathlib
import pickle
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


# ---------------------------------------------------------------------------
# checkpoints (reference: driver.py:143-189, 228-295, 343-384)
# ---------------------------------------------------------------------------
class DriverArgs(NamedTuple):
    """The driver configuration a checkpoint carries; same fields as the reference's (driver.py:228-245)."""

    setup_name: str = None
    chkpt_file: str = None
    model_parameters: dict = None
    solver_options: dict = None
    cfl_number: float = None
    end_time: float = None
    execution_mode: str = None
    fold: int = None
    resolution: int = None
    num_patches: int = None
    events: dict = dict()
    new_timestep_cadence: int = None
    verbose_output: str = ""


class DriverState(NamedTuple):
    """What ``write_checkpoint`` reads; same fields as the reference's (driver.py:279-295)."""

    iteration: int
    driver: DriverArgs
    mesh: object
    timeseries: list
    event_states: list
    solver: object
    setup: object
    cfl_number: float
    timestep_dt: float


CHECKPOINT_KEYS = ("iteration", "time", "timestep_dt", "cfl_number", "solution", "primitive", "timeseries", "solver",
                   "solver_options", "event_states", "driver", "model_parameters", "setup_name", "mesh")


def checkpoint_dict(state):
    """
    The dict the reference pickles (driver.py:158-174), key for key.  ``state.solver.solution`` is the
    ``(ni, nj, NCONS)`` float64 host array without guards; in a multi-rank run gathering it is a collective
    call, the array exists on rank 0 only and the other ranks get None back from here.
    """
    solution = state.solver.solution
    primitive = state.solver.primitive
    if solution is None:
        return None
    return dict(
        iteration=state.iteration,
        time=state.solver.time,
        timestep_dt=state.timestep_dt,
        cfl_number=state.cfl_number,
        solution=solution,
        primitive=primitive,
        timeseries=state.timeseries,
        solver=state.setup.solver,
        solver_options=state.solver.options,
        event_states=state.event_states,
        driver=state.driver,
        model_parameters=state.setup.model_parameter_dict(),
        setup_name=state.setup.dash_case_class_name(),
        mesh=state.mesh,
        **state.setup.checkpoint_diagnostics(state.solver.time),
    )


def write_checkpoint(number, outdir, state):
    """
    Write the simulation state to ``chkpt.NNNN.pk`` / ``chkpt.<name>.pk`` under `outdir` as a pickle of
    ``checkpoint_dict(state)`` (reference: driver.py:143-178).  Returns the file name (None on ranks > 0).
    """
    if type(number) is int:
        filename = f"chkpt.{number:04d}.pk"
    elif type(number) is str:
        filename = f"chkpt.{number}.pk"
    else:
        raise ValueError("number arg must be int or str")
    chkpt = checkpoint_dict(state)  # collective in a multi-rank run: every rank must come through here
    if chkpt is None:
        return None
    if outdir is not None:
        pathlib.Path(outdir).mkdir(parents=True, exist_ok=True)
        filename = os.path.join(outdir, filename)
    with open(filename, "wb") as f:
        pickle.dump(chkpt, f)
    return filename


class Recurrence(NamedTuple):
    """An event's recurrence rule as a checkpoint's ``driver.events`` holds it (reference: event.py:11-23; kind 0
    linear, 1 log).  Data only: the event loop is the reference driver's, out of scope here."""

    kind: int = None
    interval: float = None


class RecurringEvent(NamedTuple):
    """An event's state as a checkpoint's ``event_states`` holds it (reference: event.py:51-61)."""

    last_time: float = None
    number: int = 0


class _ForeignRecord(tuple):
    """Stand-in for a ``sailfish.*`` class this package has no counterpart of: keeps the constructor arguments (and
    whatever state the pickle sets).  A class, because NamedTuples pickle with NEWOBJ, which needs a type."""

    def __new__(cls, *args, **kwargs):
        self = super().__new__(cls, args)
        self.__dict__.update(kwargs)
        return self


# What a checkpoint may name.  Nothing else is resolved: a checkpoint is data, and unpickling must not be a way to
# reach os.system or eval through a GLOBAL opcode (newest_chkpt_in_directory opens every candidate in a directory).
_CHECKPOINT_MODULES = ("sailfish_amd.mesh", "sailfish_amd.driver", "sailfish_amd.physics.circumbinary",
                       "sailfish_amd.physics.kepler", "sailfish_amd.setups")
_CHECKPOINT_GLOBALS = {
    ("builtins", n) for n in ("complex", "set", "frozenset", "dict", "list", "tuple", "bytearray", "bytes", "float", "int",
                              "bool", "slice", "range", "str")
} | {("_codecs", "encode"), ("copyreg", "_reconstructor"), ("builtins", "object")  # (how protocols 0-2 spell bytes and plain objects)
} | {("collections", "OrderedDict"), ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy", "float64"), ("numpy", "int64"),
     ("numpy", "bool_"), ("numpy", "complex128")} | {
    (m, n) for m in ("numpy.core.multiarray", "numpy._core.multiarray") for n in ("_reconstruct", "scalar")
} | {(m, "_frombuffer") for m in ("numpy.core.numeric", "numpy._core.numeric")}


class _CheckpointUnpickler(pickle.Unpickler):
    """
    Checkpoints pickle application objects (mesh, driver arguments, point masses, enums, event tuples) and numpy
    arrays.  One written by the reference names its classes ``sailfish.<module>.<Class>``; the same classes live here
    under ``sailfish_amd`` with the same fields, so such names are resolved to ours -- a reference checkpoint restarts
    here without the reference installed; a ``sailfish.*`` class without a counterpart becomes a ``_ForeignRecord``.
    Every other global must be on the allow-list above (classes DEFINED in this package's data modules, numpy's array
    reconstructors, a few builtins); anything else is an ``UnpicklingError``.
    """

    def find_class(self, module, name):
        foreign = module == "sailfish" or module.startswith("sailfish.")
        local = "sailfish_amd" + module[len("sailfish"):] if foreign else module
        if local == "sailfish_amd.event" and name in ("Recurrence", "RecurringEvent"):
            return globals()[name]
        if local in _CHECKPOINT_MODULES and "." not in name:
            import importlib

            try:
                obj = getattr(importlib.import_module(local), name)
            except (ImportError, AttributeError):
                obj = None
            # classes defined in that module only: not what the module happens to have imported
            if isinstance(obj, type) and obj.__module__ == local:
                return obj
        if foreign:
            return _ForeignRecord
        if (module, name) in _CHECKPOINT_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"checkpoint names {module}.{name}, which no checkpoint of this package or of the "
                                     "reference holds: refusing to resolve it")


def load_checkpoint(chkpt_file):
    """Load a checkpoint dict written by ``write_checkpoint`` here or by the reference (driver.py:181-189)."""
    try:
        with open(chkpt_file, "rb") as f:
            return _CheckpointUnpickler(f).load()
    except FileNotFoundError:
        raise ValueError(f"could not open checkpoint file {chkpt_file}")


def newest_chkpt_in_directory(directory_name):
    """The highest-numbered ``chkpt.NNNN.pk`` in a directory that loads (reference: driver.py:192-209)."""
    import re

    expr = re.compile(r"chkpt\.([0-9]+)\.pk")
    matches = sorted(filter(None, (expr.search(f) for f in os.listdir(directory_name))), key=lambda m: int(m.group(1)))
    for match in reversed(matches):
        path = os.path.join(directory_name, match.group())
        try:
            load_checkpoint(path)
            return path
        except Exception:
            continue
    raise ValueError("the specified directory did not have a usable checkpoint")


def restart(chkpt, mode="hip", num_patches=None, **solver_kwargs):
    """
    Rebuild setup, mesh and solver from a checkpoint dict (or file name) as the reference's driver does
    (driver.py:343-384, 439-449): model parameters and solver options from the checkpoint, the mesh from the
    checkpointed resolution, the state through the solver's ``solution=`` argument.  Returns a ``DriverState``
    whose iteration, cfl number and dt continue the run bit for bit.
    """
    if not isinstance(chkpt, dict):
        chkpt = load_checkpoint(chkpt)
    driver = chkpt["driver"]
    setup = make_setup(chkpt["setup_name"], **chkpt["model_parameters"])
    mesh = chkpt["mesh"]
    if num_patches is None and driver is not None:
        num_patches = getattr(driver, "num_patches", None)
    solver = make_solver(
        chkpt["solver"], setup.physics, dict(chkpt["solver_options"]),
        setup=setup, mesh=mesh, time=chkpt["time"], solution=chkpt["solution"],
        num_patches=num_patches, mode=mode, **solver_kwargs,
    )
    return DriverState(
        iteration=chkpt["iteration"], driver=driver, mesh=mesh, timeseries=chkpt["timeseries"],
        event_states=chkpt["event_states"], solver=solver, setup=setup, cfl_number=chkpt["cfl_number"],
        timestep_dt=chkpt.get("timestep_dt"),
    )
