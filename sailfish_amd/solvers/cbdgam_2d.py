"""
Gamma-law (energy-conserving) circumbinary-disk solver, ``hip`` execution mode.

Public surface follows the reference's ``sailfish/solvers/cbdgam_2d.py``:
``Options`` (:23-27), ``Patch`` (:46-223), ``Solver`` (:226-520, RK2 only
:473-476).  Kernels: ``sailfish_amd/csrc/sf_gam.hpp``.
"""
from typing import NamedTuple

from ..physics.circumbinary import Diagnostic, EquationOfState, Physics, ViscosityModel
from .cbd_common import NG, PatchBase, SolverBase


class Options(NamedTuple):
    """cbdgam_2d.py:23-27"""

    pressure_floor: float = 1e-12
    density_floor: float = 1e-10
    velocity_ceiling: float = 1e16
    mach_ceiling: float = 1e5


def initial_condition(setup, mesh, time, i0=0, i1=None):
    return setup.primitive_array(time, mesh, i0, i1, 4)


class Patch(PatchBase):
    num_fields = 4

    def __init__(self, *args, buffer_outer_radius=0.0, buffer_surface_density=0.0,
                 buffer_surface_pressure=0.0, **kwargs):
        super().__init__(*args, **kwargs)
        self.buffer_outer_radius = buffer_outer_radius
        self.buffer_surface_density = buffer_surface_density
        self.buffer_surface_pressure = buffer_surface_pressure

    def _extent(self):
        return (self.xl, self.xr, self.yl, self.yr)

    def _head_args(self, m1, m2):
        return (
            self.physics.gamma_law_index,
            self.buffer_surface_density,
            self.buffer_surface_pressure,
            m1.mass + m2.mass,  # cbdgam_2d.py:164
            self.physics.buffer_driving_rate,
            self.buffer_outer_radius,
            self.physics.buffer_onset_width,
            int(self.physics.buffer_is_enabled),
        )

    def _tail_args(self, rk_param, dt):
        return (
            self.physics.alpha,
            rk_param,
            dt,
            self.options.velocity_ceiling,
            self.physics.cooling_coefficient,
            self.options.mach_ceiling,
            self.options.density_floor,
            self.options.pressure_floor,
            int(self.physics.constant_softening),
        )

    def point_mass_source_term(self, which_mass, gravity=False, accretion=False):
        """cbdgam_2d.py:104-143"""
        if which_mass not in (1, 2):
            raise ValueError("the mass must be either 1 or 2")
        m1, m2 = self.physics.point_masses(self.time)
        with self.execution_context:
            cons_rate = self.xp.zeros_like(self.primitive1)
            scaled = []
            for m in (m1, m2):
                scaled += [m.position_x, m.position_y, m.velocity_x, m.velocity_y, m.mass * gravity,
                           m.softening_length, m.sink_rate * accretion, m.sink_radius, m.sink_model.value]
            self.lib.cbdgam_2d_point_mass_source_term[self.shape](
                *self._extent(), *scaled, which_mass, self.primitive1, cons_rate,
                int(self.physics.constant_softening), self.physics.gamma_law_index,
            )
            return cons_rate[NG:-NG, NG:-NG]

    def recompute_conserved(self):
        self.lib.cbdgam_2d_primitive_to_conserved[self.shape](
            self.primitive1, self.conserved0, self.physics.gamma_law_index
        )

    def advance_rk(self, rk_param, dt, masses=None):
        m1, m2 = masses or self.physics.point_masses(self.time)
        self.lib.cbdgam_2d_advance_rk[self.shape](
            *self._extent(), self.conserved0, self.primitive1, self.primitive2,
            *self._head_args(m1, m2), *m1.kernel_args(), *m2.kernel_args(),
            *self._tail_args(rk_param, dt),
        )
        self.finish_stage(rk_param, dt)
        self.wavespeed_valid = False

    def maximum_wavespeed(self):
        with self.execution_context:
            self.launch_maximum_wavespeed(None)
            return self.wavespeed_max

    @property
    def lib_reductions(self):
        return self.lib.sfhip_cbdgam_2d_reductions

    def _reductions_args(self):
        return (int(self.physics.constant_softening), self.physics.gamma_law_index)

    def launch_maximum_wavespeed(self, masses):
        gamma = self.physics.gamma_law_index
        if self.fused:
            self.lib.sfhip_cbdgam_2d_max_wavespeed[self.shape](self.primitive1, gamma, self.wavespeed_max)
        else:
            self.lib.cbdgam_2d_wavespeed[self.shape](self.primitive1, self.wavespeeds, gamma)
            self.lib.sfhip_max_nonneg[self.wavespeeds.numel()](self.wavespeeds, self.wavespeed_max)

    def launch_stage(self, rk_param, dt, rd, un, wr, masses, next_masses, i_begin, i_end, want_wavespeed,
                     wavespeed_floor=0.0, defer=False, on_side_stream=False, amax_device=None, dt_dx=0.0,
                     dt_cfl=0.0, zero_slot=None, fold_into=None, host_seq=0.0):
        """One fused stage launch (sfhip_cbdgam_2d_stage); see the cbdiso_2d Patch."""
        m1, m2 = masses
        wmax = self.wavespeed_max if fold_into is None else fold_into
        if self.plan is not None:
            fire = self.plan.stage(self, m1, m2, None, None, un, rd, wr, rk_param, dt, i_begin, i_end, want_wavespeed,
                                   wavespeed_floor, wmax, amax_device, dt_dx, dt_cfl, zero_slot, on_side_stream,
                                   host_seq)
        else:
            assert host_seq == 0.0, "the pinned-memory hand-over is a service of the launch plans"
            fire = self.lib.sfhip_cbdgam_2d_stage[self.shape].bind(
                *self._extent(), un, rd, wr,
                *self._head_args(m1, m2), *m1.kernel_args(), *m2.kernel_args(),
                *self._tail_args(rk_param, dt),
                2 if un is not None else 0, int(self.first), int(self.last), i_begin, i_end, want_wavespeed,
                wavespeed_floor, wmax, None, amax_device, dt_dx, dt_cfl, zero_slot,
            )
        if defer:
            return fire
        fire()

    def plan_constants(self):
        """Constant arguments of sfhip_plan_create_cbdgam_2d, in its order (include/sfhip.h)."""
        ph, o = self.physics, self.options
        return (*self.shape, *self._extent(), ph.gamma_law_index, self.buffer_surface_density,
                self.buffer_surface_pressure, ph.buffer_driving_rate, self.buffer_outer_radius,
                ph.buffer_onset_width, int(ph.buffer_is_enabled), ph.alpha, o.velocity_ceiling,
                ph.cooling_coefficient, o.mach_ceiling, o.density_floor, o.pressure_floor,
                int(ph.constant_softening), int(self.first), int(self.last))


class Solver(SolverBase):
    """Constructor signature and checks as the reference's (cbdgam_2d.py:231-264)."""

    num_cons = 4
    patch_class = Patch
    default_rk_order = 2

    def __init__(self, setup=None, mesh=None, time=0.0, solution=None, num_patches=None, mode="hip",
                 physics=dict(), options=dict(), fused=True, backend=None, distributed=None):
        physics = dict(physics)
        physics["diagnostics"] = [
            d if isinstance(d, Diagnostic) else Diagnostic(**d) for d in physics.get("diagnostics", [])
        ]
        self._physics = physics = Physics(**physics)
        self._options = Options(**options)

        if physics.viscosity_model not in (ViscosityModel.NONE, ViscosityModel.CONSTANT_ALPHA):
            raise ValueError("solver only supports constant-alpha viscosity")
        if physics.eos_type != EquationOfState.GAMMA_LAW:
            raise ValueError("solver only supports isothermal equation of states")

        if physics.buffer_is_enabled:
            sample = [0.0] * 4  # cbdgam_2d.py:293-303
            self._buffer_outer_radius = mesh.x1
            setup.primitive(time, [mesh.x1 - physics.buffer_onset_width, 0.0], sample)
            self._buffer_surface_density = sample[0]
            self._buffer_surface_pressure = sample[3]
        else:
            self._buffer_outer_radius = 0.0
            self._buffer_surface_density = 0.0
            self._buffer_surface_pressure = 0.0

        self._setup_patches(setup, mesh, time, solution, num_patches, mode, backend, fused, distributed,
                            initial_condition)

    def _patch_kwargs(self):
        return dict(
            buffer_outer_radius=self._buffer_outer_radius,
            buffer_surface_density=self._buffer_surface_density,
            buffer_surface_pressure=self._buffer_surface_pressure,
        )

    @property
    def recommended_cfl(self):
        return 0.1
