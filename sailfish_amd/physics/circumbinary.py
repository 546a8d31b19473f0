"""
Parameter structs shared by the circumbinary solvers: the host-side source of
every scalar the kernels receive (reference: ``sailfish/physics/circumbinary.py``).
Enum values are the integers the C kernels switch on.
"""
from enum import Enum
from typing import Callable, List, NamedTuple, Union


class SinkModel(Enum):
    INACTIVE = 0
    ACCELERATION_FREE = 1
    TORQUE_FREE = 2
    FORCE_FREE = 3


class EquationOfState(Enum):
    GLOBALLY_ISOTHERMAL = 1
    LOCALLY_ISOTHERMAL = 2
    GAMMA_LAW = 3


class ViscosityModel(Enum):
    NONE = 0
    CONSTANT_NU = 1
    CONSTANT_ALPHA = 2


class Diagnostic(NamedTuple):
    quantity: str
    gravity: bool = False
    accretion: bool = False
    which_mass: Union[int, str] = None
    radial_cut: tuple = None


class PointMass(NamedTuple):
    """A softened gravitating particle that may also act as a sink (circumbinary.py:41-95)."""

    mass: float = 0.0
    position_x: float = 0.0
    position_y: float = 0.0
    velocity_x: float = 0.0
    velocity_y: float = 0.0
    softening_length: float = 0.0
    sink_model: SinkModel = SinkModel.INACTIVE
    sink_rate: float = 0.0
    sink_radius: float = 0.0

    def kernel_args(self):
        """The nine scalars in the order every kernel signature lists them (cbdiso_2d.c:401-409)."""
        return (
            self.position_x, self.position_y, self.velocity_x, self.velocity_y,
            self.mass, self.softening_length, self.sink_rate, self.sink_radius,
            self.sink_model.value,
        )


class Physics(NamedTuple):
    """Physics configuration of cbdiso_2d / cbdgam_2d (circumbinary.py:98-226); same fields and defaults."""

    eos_type: EquationOfState = EquationOfState.GLOBALLY_ISOTHERMAL
    sound_speed: float = 1.0
    mach_number: float = 10.0
    gamma_law_index: float = 5.0 / 3.0
    viscosity_model: ViscosityModel = ViscosityModel.NONE
    viscosity_coefficient: float = 0.0
    alpha: float = 0.0
    buffer_is_enabled: bool = False
    buffer_driving_rate: float = 1000.0
    buffer_onset_width: float = 0.1
    point_mass_function: Callable[[float], List[PointMass]] = None
    cooling_coefficient: float = 0.0
    constant_softening: bool = True
    diagnostics: List[Diagnostic] = []

    @property
    def num_particles(self):
        return 0 if self.point_mass_function is None else len(self.point_mass_function(0.0))

    def point_masses(self, time):
        """Always two masses; missing ones are inert defaults (circumbinary.py:202-226)."""
        masses = None if self.point_mass_function is None else self.point_mass_function(time)
        if masses is None:
            return PointMass(), PointMass()
        if isinstance(masses, PointMass):
            return masses, PointMass()
        if isinstance(masses, (tuple, list)):
            if len(masses) == 1:
                return masses[0], PointMass()
            if len(masses) == 2:
                return tuple(masses)
        raise ValueError("point_mass_function returned an unsupported description of point masses")
