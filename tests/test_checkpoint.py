"""
The checkpoint layout (reference: sailfish/driver.py:143-189 write/load, :343-384 and
:439-449 restart): a pickled dict whose keys, value types and array layout are the reference's, so that its
post-processing scripts (scripts/upsample.py, to_kitp_snapshot.py, plot.py read "solution", "mesh", "driver",
"time", "solver", "primitive", "point_masses", "model_parameters") and its restart logic find what they expect.
A restart from the file continues the run bit for bit.
"""
import os
import pickle

import numpy as np
import pytest

from sailfish_amd.driver import (CHECKPOINT_KEYS, DriverArgs, DriverState, Recurrence, RecurringEvent, load_checkpoint,
                                 newest_chkpt_in_directory, restart, timestep, write_checkpoint)
from sailfish_amd.mesh import PlanarCartesian2DMesh
from sailfish_amd.physics.circumbinary import PointMass
from sailfish_amd.setups import make_setup
from sailfish_amd.solvers import make_solver

# the keys of the reference's state_checkpoint_dict, driver.py:158-173, in its order
REFERENCE_KEYS = ["iteration", "time", "timestep_dt", "cfl_number", "solution", "primitive", "timeseries", "solver",
                  "solver_options", "event_states", "driver", "model_parameters", "setup_name", "mesh"]


def run_and_checkpoint(tmp_path, model, how, steps=3, num_patches=2, resolution=48):
    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(resolution)
    solver = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, num_patches=num_patches, **how)
    cfl = solver.recommended_cfl
    dt = None
    for _ in range(steps):
        dt = solver.step(cfl)
    driver = DriverArgs(setup_name="circumbinary-disk", model_parameters=dict(model), solver_options={}, cfl_number=cfl,
                        execution_mode="hip", fold=10, resolution=resolution, num_patches=num_patches,
                        new_timestep_cadence=1)
    state = DriverState(iteration=steps, driver=driver, mesh=mesh, timeseries=[[0.0, 1.0]], event_states=[], solver=solver,
                        setup=setup, cfl_number=cfl, timestep_dt=dt)
    return state, write_checkpoint(steps, str(tmp_path), state)


def check_layout_and_restart(tmp_path, model, how):
    state, path = run_and_checkpoint(tmp_path, model, how)
    assert path.endswith("chkpt.0003.pk") and newest_chkpt_in_directory(str(tmp_path)) == path
    chkpt = load_checkpoint(path)
    assert list(chkpt)[:len(REFERENCE_KEYS)] == REFERENCE_KEYS == list(CHECKPOINT_KEYS)
    assert list(chkpt)[len(REFERENCE_KEYS):] == ["point_masses"]  # CircumbinaryDisk.checkpoint_diagnostics
    nq = 3 if state.setup.solver == "cbdiso_2d" else 4
    assert chkpt["solution"].shape == (48, 48, nq) and chkpt["solution"].dtype == np.float64
    assert chkpt["primitive"] is None  # as the reference's cbd solvers: the solution array is primitive already
    assert chkpt["solver"] == state.setup.solver and chkpt["setup_name"] == "circumbinary-disk"
    assert chkpt["iteration"] == 3 and chkpt["time"] == state.solver.time and chkpt["timestep_dt"] == state.timestep_dt
    assert isinstance(chkpt["mesh"], PlanarCartesian2DMesh) and chkpt["mesh"] == state.mesh
    assert isinstance(chkpt["driver"], DriverArgs) and chkpt["driver"].resolution == 48
    assert chkpt["driver"]._replace(resolution=96).resolution == 96          # scripts/upsample.py:13
    assert chkpt["mesh"]._replace(ni=96, nj=96).shape == (96, 96)            # scripts/upsample.py:14
    assert all(isinstance(m, PointMass) for m in chkpt["point_masses"]) and len(chkpt["point_masses"]) == 2
    assert chkpt["solver_options"] == state.solver.options
    assert chkpt["model_parameters"]["mach_number"] == 10.0 and set(model.items()) <= set(chkpt["model_parameters"].items())

    # restart (driver.py:343-384, 439-449) continues bit for bit, also on a different patch count
    expect = [state.solver.step(state.cfl_number) for _ in range(2)]  # the uninterrupted run goes on
    final, t_final = state.solver.solution, state.solver.time
    for num_patches in (2, 1):
        again = restart(path, num_patches=num_patches, **how)
        assert again.iteration == 3 and again.solver.time == chkpt["time"]
        assert np.array_equal(again.solver.solution, chkpt["solution"])
        dts = [again.solver.step(again.cfl_number) for _ in range(2)]
        assert dts == expect and again.solver.time == t_final
        assert np.array_equal(again.solver.solution, final)
    return path


@pytest.mark.parametrize("model", [{}, dict(eos="gamma-law")], ids=["iso", "gam"])
def test_checkpoint_layout_and_bitwise_restart_cpu(tmp_path, model):
    from oracle_backend import OracleBackend

    check_layout_and_restart(tmp_path, model, dict(backend=OracleBackend(), distributed=False))


@pytest.mark.gpu
@pytest.mark.parametrize("model", [{}, dict(eos="gamma-law")], ids=["iso", "gam"])
def test_checkpoint_layout_and_bitwise_restart_gpu(tmp_path, model):
    check_layout_and_restart(tmp_path, model, dict(mode="hip"))


def test_reference_written_checkpoint_loads_without_the_reference(tmp_path):
    """A checkpoint written by the reference names its classes sailfish.mesh.PlanarCartesian2DMesh,
    sailfish.driver.DriverArgs, sailfish.physics.circumbinary.PointMass / SinkModel, ...; the loader resolves
    them to this package's classes of the same fields.  (Emulated by rewriting the class paths of a protocol-2
    pickle, whose GLOBAL opcodes are plain text: no reference-written file exists to load, and none ships.)"""
    from oracle_backend import OracleBackend

    how = dict(backend=OracleBackend(), distributed=False)
    state, path = run_and_checkpoint(tmp_path, {}, how)
    ours = load_checkpoint(path)
    blob = pickle.dumps(ours, protocol=2)
    assert b"csailfish_amd." in blob
    foreign = tmp_path / "chkpt.0099.pk"
    foreign.write_bytes(blob.replace(b"csailfish_amd.", b"csailfish."))
    with pytest.raises(ModuleNotFoundError):
        pickle.loads(foreign.read_bytes())  # the plain unpickler would need the reference installed
    chkpt = load_checkpoint(str(foreign))
    assert isinstance(chkpt["mesh"], PlanarCartesian2DMesh) and isinstance(chkpt["driver"], DriverArgs)
    assert isinstance(chkpt["point_masses"][0], PointMass) and chkpt["point_masses"] == ours["point_masses"]
    assert np.array_equal(chkpt["solution"], ours["solution"])
    again = restart(chkpt, **how)
    assert again.solver.time == state.solver.time
    assert again.solver.step(again.cfl_number) == state.solver.step(state.cfl_number)
    assert np.array_equal(again.solver.solution, state.solver.solution)


def test_reference_checkpoint_with_events_and_event_states(tmp_path):
    """What the reference's checkpoint EVENT writes: driver.events holds Recurrence rules and event_states holds
    RecurringEvent tuples (event.py:11, 51) -- NamedTuples, pickled with NEWOBJ at the default protocol, so whatever
    stands in for them must be a type.  A sailfish.* class without a counterpart here loads as a record of its
    constructor arguments."""
    import collections
    import sys
    import types

    from sailfish_amd.driver import Recurrence, RecurringEvent

    # the reference's two tuples as its pickles name them: a throw-away module `sailfish.event` (and an unknown class)
    fake = types.ModuleType("sailfish.event")
    for cls_name, fields in (("Recurrence", "kind interval"), ("RecurringEvent", "last_time number"), ("Mystery", "a b")):
        cls = collections.namedtuple(cls_name, fields)
        cls.__module__ = "sailfish.event"
        setattr(fake, cls_name, cls)
    sys.modules["sailfish.event"] = fake
    sys.modules.setdefault("sailfish", types.ModuleType("sailfish"))
    try:
        payload = dict(event_states={"checkpoint": fake.RecurringEvent(3.0, 4), "timeseries": fake.RecurringEvent(None, 0)},
                       events={"checkpoint": fake.Recurrence(0, 1.0), "timeseries": fake.Recurrence(1, 0.01)},
                       extra=fake.Mystery(1, "two"), solution=np.arange(6.0).reshape(2, 3), z=complex(1, 2))
        path = tmp_path / "chkpt.0003.pk"
        path.write_bytes(pickle.dumps(payload))  # default protocol: NEWOBJ for the tuples
    finally:
        del sys.modules["sailfish.event"]
        if not hasattr(sys.modules.get("sailfish"), "__path__"):
            sys.modules.pop("sailfish", None)
    chkpt = load_checkpoint(str(path))
    assert chkpt["event_states"]["checkpoint"] == RecurringEvent(3.0, 4)
    assert isinstance(chkpt["event_states"]["checkpoint"], RecurringEvent) and chkpt["event_states"]["checkpoint"].number == 4
    assert chkpt["events"]["timeseries"] == Recurrence(1, 0.01) and isinstance(chkpt["events"]["checkpoint"], Recurrence)
    assert tuple(chkpt["extra"]) == (1, "two")
    assert np.array_equal(chkpt["solution"], np.arange(6.0).reshape(2, 3)) and chkpt["z"] == complex(1, 2)


def test_checkpoint_loader_resolves_nothing_but_checkpoint_classes(tmp_path):
    """The loader is for data: a pickle that names anything outside the allow-list (os.system, builtins.eval, a
    module attribute that is not a class of this package's data modules) is refused, not executed."""
    evil = [
        b"cos\nsystem\n(S'echo pwned'\ntR.",
        b"cbuiltins\neval\n(S'1+1'\ntR.",
        b"csailfish_amd.driver\nos\n.",            # what a module imported is not a checkpoint class
        b"csailfish_amd.driver\nload_checkpoint\n.",  # nor are its functions
        b"csubprocess\nPopen\n.",
    ]
    for k, blob in enumerate(evil):
        path = tmp_path / f"chkpt.{k:04d}.pk"
        path.write_bytes(blob)
        with pytest.raises(pickle.UnpicklingError):
            load_checkpoint(str(path))
    from sailfish_amd.driver import newest_chkpt_in_directory

    with pytest.raises(ValueError):
        newest_chkpt_in_directory(str(tmp_path))  # none of them is usable, and none ran


def test_write_checkpoint_argument_checks(tmp_path):
    from oracle_backend import OracleBackend

    state, _ = run_and_checkpoint(tmp_path, {}, dict(backend=OracleBackend(), distributed=False), steps=1,
                                  num_patches=1, resolution=16)
    assert write_checkpoint("final", str(tmp_path), state).endswith("chkpt.final.pk")
    with pytest.raises(ValueError):
        write_checkpoint(1.5, str(tmp_path), state)
    with pytest.raises(ValueError):
        load_checkpoint(str(tmp_path / "missing.pk"))


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def continue_reference_checkpoint(tag, how, tol):
    """A checkpoint the REFERENCE's own driver wrote (tests/golden/ref_chkpt_<tag>.pk: its simulate() loop with a
    checkpoint and a timeseries event, its write_checkpoint; made by tests/golden/generate_checkpoint.py) loaded and
    restarted here, without the reference anywhere on the path, and stepped on to where the reference's run stood at
    its next-but-one checkpoint event (ref_chkpt_<tag>_continued.npz)."""
    import sys

    assert not any(m == "sailfish" or m.startswith("sailfish.") for m in sys.modules), "the reference must not be importable here"
    path = os.path.join(GOLDEN, f"ref_chkpt_{tag}.pk")
    with pytest.raises(ModuleNotFoundError):
        with open(path, "rb") as f:
            pickle.load(f)  # the plain unpickler would need the reference installed
    chkpt = load_checkpoint(path)
    g = np.load(os.path.join(GOLDEN, f"ref_chkpt_{tag}_continued.npz"))
    # what the reference's file holds, as this package's classes of the same fields
    assert list(chkpt)[:len(REFERENCE_KEYS)] == REFERENCE_KEYS and "point_masses" in chkpt
    assert isinstance(chkpt["mesh"], PlanarCartesian2DMesh) and chkpt["mesh"].shape == (32, 32)
    assert isinstance(chkpt["driver"], DriverArgs) and chkpt["driver"].setup_name == "circumbinary-disk"
    assert chkpt["driver"].num_patches == 2 and chkpt["driver"].fold == 5 and chkpt["driver"].execution_mode == "cpu"
    assert isinstance(chkpt["driver"].events["checkpoint"], Recurrence) and chkpt["driver"].events["checkpoint"].kind == 0
    assert isinstance(chkpt["event_states"]["checkpoint"], RecurringEvent) and chkpt["event_states"]["checkpoint"].number == 2
    assert chkpt["event_states"]["timeseries"].number >= 2 and len(chkpt["timeseries"]) >= 1
    assert all(isinstance(m, PointMass) for m in chkpt["point_masses"])
    assert chkpt["iteration"] == int(g["iteration0"]) and chkpt["time"] == float(g["time0"])
    assert chkpt["solver"] == ("cbdiso_2d" if tag == "iso" else "cbdgam_2d") and chkpt["cfl_number"] == float(g["cfl"])
    # restart as the reference's driver does (driver.py:343-384, 439-449) and run on, its loop's two lines per iteration
    state = restart(chkpt, **how)
    assert state.iteration == chkpt["iteration"] and len(state.solver.patches) == 2
    dt = None
    for _ in range(int(g["iteration"]) - state.iteration):
        dt = timestep(state.mesh, state.solver, state.cfl_number)
        state.solver.advance(dt)
    if tol == 0.0:
        assert state.solver.time == float(g["time"]) and dt == float(g["timestep_dt"])
        assert np.array_equal(state.solver.solution, g["solution"])
    else:
        assert abs(state.solver.time - float(g["time"])) <= 1e-13 * float(g["time"])
        err = np.abs(state.solver.solution - g["solution"]).max(axis=-1) / np.abs(g["solution"]).max(axis=-1)
        assert err.max() <= tol, err.max()


@pytest.mark.parametrize("tag", ["iso", "gam"])
def test_checkpoint_written_by_the_reference_restarts_here_cpu(tag):
    """Host logic + oracle kernels: the continuation equals the reference's own, bit for bit."""
    from oracle_backend import OracleBackend

    continue_reference_checkpoint(tag, dict(backend=OracleBackend(), distributed=False), 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["iso", "gam"])
def test_checkpoint_written_by_the_reference_restarts_here_gpu(tag):
    continue_reference_checkpoint(tag, dict(mode="hip"), 1e-12)
