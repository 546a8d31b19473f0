"""
Isothermal circumbinary-disk solver, third-order discontinuous Galerkin, ``hip`` execution mode.

Public surface follows the reference's ``sailfish/solvers/cbdisodg_2d.py``: ``Options`` (:23-30),
``initial_condition`` (:40-80), ``Patch`` (:83-299) and ``Solver`` (:302-548).  The arithmetic is
in libsfhip.so (``sailfish_amd/csrc/sf_dg.hpp``).

State: modal weights ``(ni + 2, nj + 2, 3, 3, 3)`` per patch -- one guard zone per side, conserved
field, Legendre orders in x and y -- resident on the device.

reference, per RK stage and patch              here (``fused=True``, the default)
---------------------------------------------  ------------------------------------------------
set_bc: ~8 strided slice copies (x2)            outflow edges are index clamps inside the kernels;
                                                only the one-row halo slabs between patches move
slope_limit, swap; advance_rk, swap             same two kernels, on rotating buffers
new_iteration: weights0[...] = weights1[...]    no copy: the step's first buffer is kept as u^n
wavespeed kernel + xp.max() per step            folded into the last stage's epilogue (exact
                                                integer atomic max), dt formed on the device

With ``fused=False`` the solver drives the reference-named entry points in the reference's own
order (copy, set_bc slice copies, slope_limit, set_bc, advance_rk, wavespeed array + max): the
drop-in path, kept for parity tests and for measuring what the fusions buy.

``Solver.reductions()`` is not provided: the reference's own version calls a kernel
(``cbdiso_2d_point_mass_source_term``) that its cbdisodg_2d library does not contain
(cbdisodg_2d.py:139, commented out at cbdisodg_2d.c:986-1105) and cannot run.
"""
from logging import getLogger
from typing import NamedTuple

import numpy as np

from ..physics.circumbinary import EquationOfState, Physics, ViscosityModel
from ..subdivide import concat_on_host, subdivide
from .cbd_common import Comm, HipBackend, SolverBase

logger = getLogger(__name__)

NCONS = 3
NPOLY = 6
ORDER = 3
GUARD = 1

# Gauss-Legendre nodes / weights and scaled Legendre polynomials at the nodes, as the reference's
# Python spells them (cbdisodg_2d.py:46-52)
GAUSS_X = (-0.774596669241483, +0.000000000000000, +0.774596669241483)
GAUSS_W = (+0.555555555555556, +0.888888888888889, +0.555555555555556)
LEGENDRE = (
    (+1.000000000000000, +1.000000000000000, +1.000000000000000),
    (-1.341640786499873, +0.000000000000000, +1.341640786499873),
    (+0.894427190999914, -1.118033988749900, +0.894427190999914),
)


class Options(NamedTuple):
    """Solver-specific options (cbdisodg_2d.py:23-30)."""

    velocity_ceiling: float = 1e12
    rk_order: int = 2
    limit_slopes: bool = True


def initial_condition(setup, mesh, time, i0=0, i1=None):
    """
    Rows [i0, i1) of the initial weights, shape (i1 - i0, nj, 3, 3, 3): L2 projection of the
    setup's conserved state by 3x3 Gauss quadrature (cbdisodg_2d.py:40-80).  Vectorised over zones;
    each zone accumulates its nine nodes in the reference's order with the reference's expression,
    so the result equals the per-zone loop bit for bit whenever ``setup.primitive_points`` does.
    """
    i1 = mesh.shape[0] if i1 is None else i1
    nj = mesh.shape[1]
    dx, dy = mesh.dx, mesh.dy
    i = np.arange(i0, i1, dtype=np.float64)[:, None]
    j = np.arange(nj, dtype=np.float64)[None, :]
    xc = mesh.x0 + (i + 0.5) * dx
    yc = mesh.y0 + (j + 0.5) * dy
    weights = np.zeros([i1 - i0, nj, NCONS, ORDER, ORDER])
    for a in range(ORDER):
        for b in range(ORDER):
            x = np.broadcast_to(xc + 0.5 * dx * GAUSS_X[a], (i1 - i0, nj))
            y = np.broadcast_to(yc + 0.5 * dy * GAUSS_X[b], (i1 - i0, nj))
            prim = setup.primitive_points(time, x, y, NCONS)
            cons = np.empty_like(prim)
            cons[..., 0] = prim[..., 0]
            cons[..., 1] = prim[..., 0] * prim[..., 1]
            cons[..., 2] = prim[..., 0] * prim[..., 2]
            for m in range(ORDER):
                for n in range(ORDER):
                    weights[..., m, n] += 0.25 * cons * LEGENDRE[m][a] * LEGENDRE[n][b] * GAUSS_W[a] * GAUSS_W[b]
    return weights


class Patch:
    """
    One slab of the mesh with its device arrays (reference: cbdisodg_2d.py:83-299).  ``weights1`` is
    the current state, ``weights0`` the state at the start of the step, ``weights2`` scratch.
    """

    def __init__(self, time, weights, mesh, index_range, physics, options, buffer_outer_radius,
                 buffer_surface_density, backend, context, fused, first, last):
        i0, i1 = index_range
        self.index_range = index_range
        self.lib = backend.lib
        self.xp = xp = backend.xp
        self.mesh = mesh
        self.execution_context = context
        self.time = self.time0 = time
        self.shape = (i1 - i0, mesh.shape[1])  # not including guard zones
        self.physics = physics
        self.options = options
        self.fused = fused
        self.first = first
        self.last = last
        self.xl, self.yl = mesh.vertex_coordinates(i0, 0)
        self.xr, self.yr = mesh.vertex_coordinates(i1, mesh.shape[1])
        self.buffer_outer_radius = buffer_outer_radius
        self.buffer_surface_density = buffer_surface_density
        self.wavespeed_valid = False
        with self.execution_context:
            self.weights1 = xp.array(weights)
            self.weights2 = xp.array(weights)
            # fused: written before it is read (stage 0 reads weights1 as u^n); reference order: copied into
            self.weights0 = xp.array(weights) if fused else xp.zeros(weights.shape)
            self.wavespeed_max = xp.zeros(1)
            if not fused:
                self.wavespeeds = xp.zeros(weights.shape[:2])
        self._u0 = None  # fused: the buffer holding u^n during a step

    # ---- argument groups ---------------------------------------------------------------------
    def _extent(self):
        return (self.xl, self.xr, self.yl, self.yr)

    def _eos_args(self):
        return (self.physics.sound_speed**2, self.physics.mach_number**2, self.physics.eos_type.value)

    def _advance_args(self, w0, w1, w2, masses, rk_param, dt):
        m1, m2 = masses
        return (
            *self._extent(), w0, w1, w2,
            self.buffer_surface_density,
            m1.mass + m2.mass,  # buffer_central_mass, cbdisodg_2d.py:240
            self.physics.buffer_driving_rate,
            self.buffer_outer_radius,
            self.physics.buffer_onset_width,
            int(self.physics.buffer_is_enabled),
            *m1.kernel_args(), *m2.kernel_args(), *self._eos_args(),
            self.physics.viscosity_coefficient, rk_param, dt, self.options.velocity_ceiling,
        )

    def stage_time(self, rk_param, dt):
        """Time after a stage; the reference's expression, literally (cbdisodg_2d.py:284)."""
        return self.time0 * rk_param + (self.time + dt) * (1.0 - rk_param)

    # ---- reference-named kernels (drop-in path) ---------------------------------------------------
    def maximum_wavespeed(self):
        """Reference form: per-zone array + max (cbdisodg_2d.py:169-205); returns a device scalar."""
        with self.execution_context:
            self.launch_maximum_wavespeed(self.physics.point_masses(self.time))
            return self.wavespeed_max

    def copy_weights1_to_weights0(self):
        with self.execution_context:
            self.weights0.copy_(self.weights1)

    def slope_limit(self, masses=None):
        m1, m2 = masses or self.physics.point_masses(self.time)
        with self.execution_context:
            self.lib.cbdisodg_2d_slope_limit[self.shape](
                *self._extent(), m1.position_x, m1.position_y, m2.position_x, m2.position_y,
                self.weights1, self.weights2,
            )
        self.weights1, self.weights2 = self.weights2, self.weights1

    def advance_rk(self, rk_param, dt, masses=None):
        masses = masses or self.physics.point_masses(self.time)
        with self.execution_context:
            self.lib.cbdisodg_2d_advance_rk[self.shape](
                *self._advance_args(self.weights0, self.weights1, self.weights2, masses, rk_param, dt))
        self.time = self.stage_time(rk_param, dt)
        self.weights1, self.weights2 = self.weights2, self.weights1
        self.wavespeed_valid = False

    def new_iteration(self):
        self.time0 = self.time
        if self.fused:
            self._u0 = self.weights1
        else:
            self.copy_weights1_to_weights0()

    @property
    def primitive(self):
        """Zone-average primitives with guards, (ni + 2, nj + 2, 3) (cbdisodg_2d.py:291-299)."""
        with self.execution_context:
            u0 = self.weights1[:, :, :, 0, 0]
            p0 = self.xp.zeros_like(u0)
            p0[..., 0] = u0[..., 0]
            p0[..., 1] = u0[..., 1] / u0[..., 0]
            p0[..., 2] = u0[..., 2] / u0[..., 0]
            return p0

    # ---- launches used by the Solver -----------------------------------------------------------
    def launch_maximum_wavespeed(self, masses):
        m1, m2 = masses
        if self.fused:
            self.lib.sfhip_cbdisodg_2d_max_wavespeed[self.shape](
                *self._extent(), *self._eos_args(),
                m1.position_x, m1.position_y, m1.mass, m1.softening_length,
                m2.position_x, m2.position_y, m2.mass, m2.softening_length,
                self.options.velocity_ceiling, self.weights1, self.wavespeed_max,
            )
        else:
            self.lib.cbdisodg_2d_wavespeed[self.shape](
                *self._extent(), *self._eos_args(), *m1.kernel_args(), *m2.kernel_args(),
                self.options.velocity_ceiling, self.weights1, self.wavespeeds,
            )
            self.lib.sfhip_max_nonneg[self.wavespeeds.numel()](self.wavespeeds, self.wavespeed_max)

    def launch_limit(self, i_begin, i_end):
        """Fused path: weights1 -> weights2 on rows [i_begin, i_end); the caller swaps afterwards."""
        self.lib.sfhip_cbdisodg_2d_limit[self.shape](
            self.weights1, self.weights2, int(self.first), int(self.last), 1, i_begin, i_end)

    def launch_stage(self, rk_param, dt, masses, next_masses, i_begin, i_end, want_wavespeed, dt_device=None,
                     defer=False):
        """Fused path: (u^n, weights1) -> weights2 on rows [i_begin, i_end); the caller swaps afterwards."""
        w1, w2 = next_masses
        fire = self.lib.sfhip_cbdisodg_2d_stage[self.shape].bind(
            *self._advance_args(self._u0, self.weights1, self.weights2, masses, rk_param, dt),
            int(self.first), int(self.last), 1, i_begin, i_end, want_wavespeed,
            w1.position_x, w1.position_y, w1.mass, w1.softening_length,
            w2.position_x, w2.position_y, w2.mass, w2.softening_length,
            self.wavespeed_max, dt_device,
        )
        if defer:
            return fire
        fire()

    def swap_after_limit(self):
        """
        weights1 <- limited state.  The unlimited state is dead from here on unless it is u^n (stage 0),
        in which case the spare buffer takes its place in the rotation.
        """
        if self.weights1 is self._u0:
            self.weights1, self.weights2, self.weights0 = self.weights2, self.weights0, self.weights1
        else:
            self.weights1, self.weights2 = self.weights2, self.weights1

    def swap_after_stage(self):
        if self.weights1 is self._u0:  # stage 0 without limiter: keep u^n, write next into the spare buffer
            self.weights1, self.weights2, self.weights0 = self.weights2, self.weights0, self.weights1
        else:
            self.weights1, self.weights2 = self.weights2, self.weights1


class Solver(SolverBase):
    """
    Adapter between the driver and the cbdisodg_2d HIP kernels; constructor signature and checks as
    the reference's (cbdisodg_2d.py:307-346), plus ``fused`` / ``backend`` / ``distributed`` keywords.
    """

    num_cons = NCONS
    patch_class = Patch

    def __init__(self, setup=None, mesh=None, time=0.0, solution=None, num_patches=None, mode="hip",
                 physics=dict(), options=dict(), fused=True, backend=None, distributed=None):
        from ..mesh import PlanarCartesian2DMesh

        physics = dict(physics)
        physics.setdefault("diagnostics", [])
        self._physics = physics = Physics(**physics)
        self._options = options = Options(**options)

        if type(mesh) is not PlanarCartesian2DMesh:
            raise ValueError("solver only supports 2D Cartesian mesh")
        if setup.boundary_condition != "outflow":
            raise ValueError("solver only supports outflow boundary condition")
        if physics.viscosity_model not in (ViscosityModel.NONE, ViscosityModel.CONSTANT_NU):
            raise ValueError("solver only supports constant-nu viscosity")
        if physics.eos_type not in (EquationOfState.GLOBALLY_ISOTHERMAL, EquationOfState.LOCALLY_ISOTHERMAL):
            raise ValueError("solver only supports isothermal equation of states")
        if physics.cooling_coefficient != 0.0:
            raise ValueError("solver does not support thermal cooling")
        if not physics.constant_softening:
            raise ValueError("solver only supports constant gravitational softening")
        if options.rk_order not in (1, 2, 3):
            raise ValueError("rk_order must be 1, 2 or 3")
        if backend is None:
            if mode != "hip":
                raise ValueError(
                    f"sailfish_amd solvers implement execution mode 'hip' only (got {mode!r}); "
                    "use the reference for cpu/omp"
                )
            backend = HipBackend()

        self.backend = backend
        self.mode = backend.mode
        self.xp = backend.xp
        self.mesh = mesh
        self.setup = setup
        self.num_guard = GUARD
        self.fused = fused
        self.comm = Comm.auto(distributed)
        ni, nj = mesh.shape

        if physics.buffer_is_enabled:
            # the buffer's target state: the initial condition sampled at the onset radius on the x-axis
            # (cbdisodg_2d.py:374-383)
            sample = [0.0] * 3
            buffer_outer_radius = mesh.x1
            setup.primitive(time, [mesh.x1 - physics.buffer_onset_width, 0.0], sample)
            buffer_surface_density = sample[0]
        else:
            buffer_outer_radius = 0.0
            buffer_surface_density = 0.0

        world = self.comm.size
        if num_patches is None or num_patches < world:
            num_patches = world
        if num_patches % world:
            raise ValueError(f"num_patches ({num_patches}) must be a multiple of the number of ranks ({world})")
        per_rank = num_patches // world
        ranges = subdivide(ni, num_patches)
        self.num_patches = num_patches
        self.patch_ranges = ranges
        self.local_ids = list(range(self.comm.rank * per_rank, (self.comm.rank + 1) * per_rank))
        self.owner = lambda k: k // per_rank

        logger.info(f"initiate with time={time:0.4f}")
        logger.info(f"subdivide grid over {num_patches} patches on {world} rank(s)")
        logger.info(f"mesh is {mesh}")
        logger.info(f"viscosity is {physics.viscosity_coefficient}")

        ndev = backend.num_devices()
        if self.comm.enabled and backend.mode == "hip":
            import os

            base_device = int(os.environ.get("LOCAL_RANK", self.comm.rank % max(ndev, 1)))
        else:
            base_device = None

        self.patches = []
        for n, k in enumerate(self.local_ids):
            a, b = ranges[k]
            weights = np.zeros([b - a + 2 * GUARD, nj + 2 * GUARD, NCONS, ORDER, ORDER])
            if solution is None:
                weights[GUARD:-GUARD, GUARD:-GUARD] = initial_condition(setup, mesh, time, a, b)
            else:
                weights[GUARD:-GUARD, GUARD:-GUARD] = solution[a:b]
            device = base_device if base_device is not None else (n % max(ndev, 1))
            self.patches.append(Patch(
                time, weights, mesh, (a, b), physics, options, buffer_outer_radius, buffer_surface_density,
                backend, backend.context(device), fused, first=(k == 0), last=(k == num_patches - 1)))
        self._comm_stream = None
        self._mass_cache = (None, None)
        self._wavespeed_floor = 0.0
        self._dt_dev = None
        self._fetch_stream = None
        self._remote_neighbors = None
        self.event_log = None
        if not fused:
            self.set_bc("weights1")  # cbdisodg_2d.py:405

    # ------------------------------------------------------------------ state
    @property
    def solution(self):
        """(ni, nj, 3, 3, 3) float64 host array without guards (cbdisodg_2d.py:407-414)."""
        local = concat_on_host([p.weights1 for p in self.patches], (GUARD, GUARD), rank=2)
        parts = self.comm.gather_arrays(local)  # collective: the array on rank 0, None on the other ranks
        return None if parts is None else np.concatenate(parts, axis=0)

    @property
    def primitive(self):
        """(ni, nj, 3) zone-average primitives on the host (cbdisodg_2d.py:416-423)."""
        local = concat_on_host([p.primitive for p in self.patches], (GUARD, GUARD), rank=2)
        parts = self.comm.gather_arrays(local)  # collective: the array on rank 0, None on the other ranks
        return None if parts is None else np.concatenate(parts, axis=0)

    def reductions(self):
        raise NotImplementedError(
            "cbdisodg_2d has no time-series reductions: the reference's own implementation calls a kernel "
            "its library does not contain (cbdisodg_2d.py:139)")

    @property
    def recommended_cfl(self):
        return 0.08

    @property
    def maximum_cfl(self):
        return 0.4

    # -------------------------------------------------------------------- dt
    def maximum_wavespeed(self):
        """
        Global max wavespeed (cbdisodg_2d.py:479-488).  After a fused step the per-patch maxima are
        already on the device (epilogue of the last stage, exact) and no kernel runs.
        """
        have = self.fused and all(p.wavespeed_valid for p in self.patches)
        return self._reduce_wavespeed(recompute=not have)

    # --------------------------------------------------------------- advance
    def new_iteration(self):
        for p in self.patches:
            p.new_iteration()

    def advance(self, dt):
        """One full Runge-Kutta step (cbdisodg_2d.py:490-500)."""
        self.new_iteration()
        params = self.rk_parameters()
        for stage, a in enumerate(params):
            self.advance_rk(a, dt, stage, len(params))

    def advance_rk(self, rk_param, dt, stage=0, nstages=1, dt_device=None):
        """One stage: optional slope limiting, then the update (cbdisodg_2d.py:502-509)."""
        if not self.fused:
            if self._options.limit_slopes:
                self.set_bc("weights1")
                for p in self.patches:
                    p.slope_limit(self.point_masses(p.time))
            self.set_bc("weights1")
            for p in self.patches:
                p.advance_rk(rk_param, dt, self.point_masses(p.time))
            return

        last_stage = stage == nstages - 1
        if self._options.limit_slopes:
            self.finish_halo_exchange(self.start_halo_exchange("weights1"))
            for p in self.patches:
                with p.execution_context:
                    self._launch_limit(p)
                p.swap_after_limit()
        self.finish_halo_exchange(self.start_halo_exchange("weights1"))
        for p in self.patches:
            masses = self.point_masses(p.time)
            next_masses = self.point_masses(p.stage_time(rk_param, dt)) if last_stage else masses
            with p.execution_context:
                self._launch_stage(p, rk_param, dt, masses, next_masses, 2 if last_stage else 0, dt_device)
            p.swap_after_stage()
            if dt_device is None:
                p.time = p.stage_time(rk_param, dt)
            p.wavespeed_valid = last_stage

    def _launch_stage(self, p, rk_param, dt, masses, next_masses, want_wavespeed, dt_device):
        ni = p.shape[0]
        fire = p.launch_stage(rk_param, dt, masses, next_masses, 0, ni, want_wavespeed, dt_device, defer=True)
        self._timed(p, fire, "dg_advance_kernel", 3 * 18 * 8.0)  # reads u^n and w, writes w': 18 used slots each

    def _launch_limit(self, p):
        fire = lambda: p.launch_limit(0, p.shape[0])  # noqa: E731
        self._timed(p, fire, "dg_limit_kernel", 2 * 27 * 8.0)  # copies all 27 slots, as the reference's does

    def _timed(self, p, fire, kernel, bytes_per_zone):
        """Run one launch, bracketed by HIP events on its stream when bench.py asked for an event log."""
        if self.event_log is None:
            fire()
            return
        torch = self._torch()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fire()
        e1.record()
        self.event_log.append(dict(zones=p.shape[0] * p.shape[1], kernel=kernel, bytes_per_zone=bytes_per_zone,
                                   start=e0, stop=e1))

    def step(self, cfl_number, dx=None):
        """
        One driver iteration -- ``dt = dx / maximum_wavespeed() * cfl; advance(dt)`` (driver.py:508-511).
        Fused, multi-stage: dt is also formed on the device from the reduced wavespeed (same IEEE
        operations, sfhip_timestep) and stage 0 is launched against that device value while the host
        reads the wavespeed back on a side stream for its own bookkeeping.  Returns dt.
        """
        dx = self.mesh.min_spacing() if dx is None else dx
        params = self.rk_parameters()
        same_device = len({str(p.wavespeed_max.device) for p in self.patches}) == 1
        if not (self.fused and len(params) > 1 and same_device and hasattr(self.patches[0].lib, "sfhip_timestep")):
            dt = dx / self.maximum_wavespeed() * cfl_number
            self.advance(dt)
            return dt
        have = all(p.wavespeed_valid for p in self.patches)
        amax_dev = self._reduce_wavespeed_device(recompute=not have)
        p0 = self.patches[0]
        if self._dt_dev is None:
            with p0.execution_context:
                self._dt_dev = p0.xp.zeros(1)
        with p0.execution_context:
            fetch = self._start_scalar_fetch(amax_dev)
            p0.lib.sfhip_timestep[1](float(dx), float(cfl_number), amax_dev, self._dt_dev)
        self.new_iteration()
        self.advance_rk(params[0], float("nan"), 0, len(params), dt_device=self._dt_dev)
        dt = dx / fetch() * cfl_number
        for p in self.patches:
            p.time = p.stage_time(params[0], dt)
        for stage in range(1, len(params)):
            self.advance_rk(params[stage], dt, stage, len(params))
        return dt
