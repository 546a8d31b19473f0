#!/usr/bin/env python3
"""
Writes tests/golden/ref_chkpt_{iso,gam}.pk and ref_chkpt_{iso,gam}_continued.npz by RUNNING THE REFERENCE's own driver.

A checkpoint is the reference's on-disk contract (driver.py:143-189); until round 3 the tests only had checkpoints
this package wrote itself, with the class paths rewritten.  Here the reference's `simulate` generator runs the
circumbinary disk with a `checkpoint` event (the reference's solvers, setups, event tuples and DriverArgs, its C kernels
built by oracle/build_ref.py; tests/golden/generate.py's RefLibrary stands in for the absent cffi build driver, nothing
else is touched), the reference's own `write_checkpoint` pickles the state at the first checkpoint after the start, and
the run goes on to the next checkpoint event, whose state is recorded as the expected continuation (the reference
restarts bit for bit: driver.py:369-386).  The .pk files are data written by this script, not files that ship with the
reference; they name the reference's classes (sailfish.driver.DriverArgs, sailfish.event.Recurrence / RecurringEvent,
sailfish.mesh.PlanarCartesian2DMesh, sailfish.physics.circumbinary.PointMass ...), which is the point.

Runs only in the build container (needs /root/reference).
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import generate  # noqa: E402  (reference_modules: the reference importable, its kernels built and bound)


def main():
    generate.reference_modules()
    import sailfish.driver as ref_driver
    from sailfish.event import Recurrence

    for tag, model, resolution, interval in (("iso", {}, 32, 0.004), ("gam", dict(eos="gamma-law"), 32, 0.002)):
        args = ref_driver.DriverArgs(
            setup_name="circumbinary-disk", model_parameters=dict(model), solver_options=dict(), resolution=resolution,
            execution_mode="cpu", num_patches=2, fold=5, end_time=1.0,
            events={"checkpoint": Recurrence(kind=0, interval=interval), "timeseries": Recurrence(kind=0, interval=interval / 2)},
        )
        tmp = tempfile.mkdtemp()
        written = None
        try:
            for name, number, state in ref_driver.simulate(args):
                if name == "timeseries":
                    state.timeseries.append(state.solver.reductions())
                if name == "checkpoint" and number == 1:
                    ref_driver.write_checkpoint(number, tmp, state)  # the reference's own writer
                    written = dict(iteration=state.iteration, time=state.solver.time, dt=state.timestep_dt)
                    shutil.copy(os.path.join(tmp, "chkpt.0001.pk"), os.path.join(HERE, f"ref_chkpt_{tag}.pk"))
                if name == "checkpoint" and number == 3:
                    assert written is not None and state.iteration > written["iteration"]
                    np.savez(os.path.join(HERE, f"ref_chkpt_{tag}_continued.npz"),
                             iteration0=written["iteration"], time0=written["time"], iteration=state.iteration,
                             time=state.solver.time, solution=state.solver.solution, cfl=state.cfl_number,
                             timestep_dt=state.timestep_dt)
                    print(tag, "checkpoint at iteration", written["iteration"], "continued to", state.iteration,
                          "t =", state.solver.time)
                    break
        finally:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
