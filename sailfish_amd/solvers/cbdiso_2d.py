"""
Isothermal circumbinary-disk solver, ``hip`` execution mode.

Public surface follows the reference's ``sailfish/solvers/cbdiso_2d.py``:
``Options`` (:23-30), ``initial_condition`` (:33-46), ``Patch`` (:49-248) and
``Solver`` (:251-590).  The arithmetic is in libsfhip.so
(``sailfish_amd/csrc/sf_iso.hpp``); see ``cbd_common`` for the data flow.
"""
from typing import NamedTuple

from ..physics.circumbinary import Diagnostic, EquationOfState, Physics, ViscosityModel
from .cbd_common import NG, PatchBase, SolverBase


class Options(NamedTuple):
    """Solver-specific options (cbdiso_2d.py:23-30)."""

    velocity_ceiling: float = 1e12
    density_floor: float = 1e-12
    rk_order: int = 2


def initial_condition(setup, mesh, time, i0=0, i1=None):
    """Rows [i0, i1) of primitive initial data, shape (i1 - i0, nj, 3)."""
    return setup.primitive_array(time, mesh, i0, i1, 3)


class Patch(PatchBase):
    num_fields = 3

    def __init__(self, *args, buffer_outer_radius=0.0, buffer_surface_density=0.0, **kwargs):
        super().__init__(*args, **kwargs)
        self.buffer_outer_radius = buffer_outer_radius
        self.buffer_surface_density = buffer_surface_density

    def _extent(self):
        return (self.xl, self.xr, self.yl, self.yr)

    def _buffer_args(self, m1, m2):
        return (
            self.buffer_surface_density,
            m1.mass + m2.mass,  # buffer_central_mass, cbdiso_2d.py:194
            self.physics.buffer_driving_rate,
            self.buffer_outer_radius,
            self.physics.buffer_onset_width,
            int(self.physics.buffer_is_enabled),
        )

    def _eos_args(self):
        return (self.physics.sound_speed**2, self.physics.mach_number**2, self.physics.eos_type.value)

    def _tail_args(self, rk_param, dt):
        return (
            *self._eos_args(),
            self.physics.viscosity_coefficient,
            rk_param,
            dt,
            self.options.velocity_ceiling,
            self.options.density_floor,
        )

    # ---- reference-named kernels (drop-in path) -----------------------------
    def point_mass_source_term(self, which_mass, gravity=False, accretion=False):
        """Rates of conserved quantities from one mass (cbdiso_2d.py:105-137)."""
        if which_mass not in (1, 2):
            raise ValueError("which_mass must be either 1 or 2")
        m = self.physics.point_masses(self.time)[which_mass - 1]
        with self.execution_context:
            cons_rate = self.xp.zeros_like(self.primitive1)
            self.lib.cbdiso_2d_point_mass_source_term[self.shape](
                *self._extent(),
                m.position_x, m.position_y, m.velocity_x, m.velocity_y,
                m.mass * gravity, m.softening_length, m.sink_rate * accretion, m.sink_radius,
                m.sink_model.value,
                self.primitive1, cons_rate,
            )
        return cons_rate[NG:-NG, NG:-NG]

    def recompute_conserved(self):
        self.lib.cbdiso_2d_primitive_to_conserved[self.shape](self.primitive1, self.conserved0)

    def advance_rk(self, rk_param, dt, masses=None):
        """Reference-order stage: cbdiso_2d_advance_rk on (conserved0, primitive1) -> primitive2."""
        m1, m2 = masses or self.physics.point_masses(self.time)
        self.lib.cbdiso_2d_advance_rk[self.shape](
            *self._extent(), self.conserved0, self.primitive1, self.primitive2,
            *self._buffer_args(m1, m2), *m1.kernel_args(), *m2.kernel_args(),
            *self._tail_args(rk_param, dt),
        )
        self.finish_stage(rk_param, dt)
        self.wavespeed_valid = False

    def maximum_wavespeed(self):
        """Reference form: per-zone array + max (cbdiso_2d.py:139-174); returns a device scalar."""
        m1, m2 = self.physics.point_masses(self.time)
        with self.execution_context:
            self.launch_maximum_wavespeed((m1, m2))
            return self.wavespeed_max

    @property
    def lib_reductions(self):
        return self.lib.sfhip_cbdiso_2d_reductions

    # ---- launches used by the Solver ---------------------------------------
    def launch_maximum_wavespeed(self, masses):
        m1, m2 = masses
        if self.fused:
            self.lib.sfhip_cbdiso_2d_max_wavespeed[self.shape](
                *self._extent(), *self._eos_args(),
                m1.position_x, m1.position_y, m1.mass, m1.softening_length,
                m2.position_x, m2.position_y, m2.mass, m2.softening_length,
                self.primitive1, self.wavespeed_max,
            )
        else:
            self.lib.cbdiso_2d_wavespeed[self.shape](
                *self._extent(), *self._eos_args(), *m1.kernel_args(), *m2.kernel_args(),
                self.primitive1, self.wavespeeds,
            )
            self.lib.sfhip_max_nonneg[self.wavespeeds.numel()](self.wavespeeds, self.wavespeed_max)

    def launch_stage(self, rk_param, dt, rd, un, wr, masses, next_masses, i_begin, i_end, want_wavespeed,
                     wavespeed_floor=0.0, defer=False, on_side_stream=False, amax_device=None, dt_dx=0.0,
                     dt_cfl=0.0, zero_slot=None, fold_into=None, host_seq=0.0):
        """
        One fused stage launch (sfhip_cbdiso_2d_stage).  `fold_into`: where want_wavespeed folds, default this
        patch's own scalar; amax_device / dt_dx / dt_cfl / zero_slot: see include/sfhip.h.  On the device the
        call goes through the patch's launch plan (constants marshalled once), elsewhere through the Library.
        """
        m1, m2 = masses
        w1, w2 = next_masses
        wmax = self.wavespeed_max if fold_into is None else fold_into
        if self.plan is not None:
            fire = self.plan.stage(self, m1, m2, w1, w2, un, rd, wr, rk_param, dt, i_begin, i_end, want_wavespeed,
                                   wavespeed_floor, wmax, amax_device, dt_dx, dt_cfl, zero_slot, on_side_stream,
                                   host_seq)
        else:
            assert host_seq == 0.0, "the pinned-memory hand-over is a service of the launch plans"
            fire = self.lib.sfhip_cbdiso_2d_stage[self.shape].bind(
                *self._extent(), un, rd, wr,
                *self._buffer_args(m1, m2), *m1.kernel_args(), *m2.kernel_args(),
                *self._tail_args(rk_param, dt),
                2 if un is not None else 0, int(self.first), int(self.last), i_begin, i_end, want_wavespeed,
                wavespeed_floor,
                w1.position_x, w1.position_y, w1.mass, w1.softening_length,
                w2.position_x, w2.position_y, w2.mass, w2.softening_length,
                wmax, None, amax_device, dt_dx, dt_cfl, zero_slot,
            )
        if defer:
            return fire
        fire()

    def plan_constants(self):
        """Constant arguments of sfhip_plan_create_cbdiso_2d, in its order (include/sfhip.h)."""
        ph, o = self.physics, self.options
        return (*self.shape, *self._extent(), self.buffer_surface_density, ph.buffer_driving_rate,
                self.buffer_outer_radius, ph.buffer_onset_width, int(ph.buffer_is_enabled), *self._eos_args(),
                ph.viscosity_coefficient, o.velocity_ceiling, o.density_floor, int(self.first), int(self.last))


class Solver(SolverBase):
    """
    Adapter between the driver and the cbdiso_2d HIP kernels; constructor
    signature and checks as the reference's (cbdiso_2d.py:256-298), plus
    ``fused`` / ``backend`` / ``distributed`` keywords.
    """

    num_cons = 3
    patch_class = Patch

    def __init__(self, setup=None, mesh=None, time=0.0, solution=None, num_patches=None, mode="hip",
                 physics=dict(), options=dict(), fused=True, backend=None, distributed=None):
        physics = dict(physics)
        physics["diagnostics"] = [
            d if isinstance(d, Diagnostic) else Diagnostic(**d) for d in physics.get("diagnostics", [])
        ]
        self._physics = physics = Physics(**physics)
        self._options = options = Options(**options)

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

        if physics.buffer_is_enabled:
            # the buffer's target state is the initial condition sampled at the
            # onset radius on the x-axis (cbdiso_2d.py:325-337)
            sample = [0.0] * 3
            self._buffer_outer_radius = mesh.x1
            setup.primitive(time, [mesh.x1 - physics.buffer_onset_width, 0.0], sample)
            self._buffer_surface_density = sample[0]
        else:
            self._buffer_outer_radius = 0.0
            self._buffer_surface_density = 0.0

        self._setup_patches(setup, mesh, time, solution, num_patches, mode, backend, fused, distributed,
                            initial_condition)

    def _patch_kwargs(self):
        return dict(
            buffer_outer_radius=self._buffer_outer_radius,
            buffer_surface_density=self._buffer_surface_density,
        )

    @property
    def recommended_cfl(self):
        return 0.3
