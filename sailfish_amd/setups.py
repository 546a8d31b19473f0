"""
The setups that feed the hot path.

``CircumbinaryDisk`` is the benchmark problem (reference:
``sailfish/setups/circumbinary_disk.py:17-215``): it defines the synthetic input
of every BASELINE.json config (initial data per point, physics dict, point-mass
callback).  ``CylindricalExplosion`` (``sailfish/setups/simple2d.py:40-114``) is
the minimal-physics second parity input.  The reference's ``SetupBase``
machinery (model-parameter parsing, registry, CLI help) is configuration only
and is not rebuilt; parameters are plain constructor keywords here.

Besides the reference's per-point ``primitive(t, coords, out)`` both classes
offer ``primitive_array(t, mesh, i0, i1)``: the same initial data for whole rows
at once.  The reference fills the array with a Python double loop
(cbdiso_2d.py:33-46), which is 2.7e8 interpreter calls at 16384^2.
"""
from math import exp, pi, sqrt

import numpy as np

from .mesh import LogSphericalMesh, PlanarCartesian2DMesh
from .physics.circumbinary import EquationOfState, PointMass, SinkModel, ViscosityModel
from .physics.kepler import OrbitalElements


class SetupError(Exception):
    pass


class Setup:
    """Minimal stand-in for the attributes of SetupBase the solvers and driver read."""

    start_time = 0.0
    reference_time_scale = 1.0
    default_end_time = None
    boundary_condition = "outflow"

    def __init__(self, **kwargs):
        for key, value in kwargs.items():
            if not hasattr(type(self), key):
                raise SetupError(f"{type(self).__name__} has no model parameter {key}")
            setattr(self, key, value)
        self.validate()

    def validate(self):
        pass

    # -- what a checkpoint records about the setup (setup_base.py:221-286) ---------------------------------
    @classmethod
    def dash_case_class_name(cls):
        """The name the setup is registered and checkpointed under, e.g. "circumbinary-disk"."""
        for name, setup_class in SETUPS.items():
            if setup_class is cls:
                return name
        raise SetupError(f"{cls.__name__} is not a registered setup")

    def model_parameter_dict(self):
        """Every model parameter with its current value (class-level defaults overridden by the constructor)."""
        out = {}
        for klass in reversed(type(self).__mro__):
            for key, value in vars(klass).items():
                if key.startswith("_") or callable(value) or isinstance(value, (property, classmethod, staticmethod)):
                    continue
                if key in ("start_time", "reference_time_scale", "default_end_time", "default_resolution",
                           "boundary_condition"):
                    continue
                out[key] = getattr(self, key)
        return out

    def checkpoint_diagnostics(self, time):
        """Post-processing data a checkpoint carries besides the state: the point masses, where the setup has
        them (circumbinary_disk.py:214-215)."""
        if hasattr(self, "point_masses"):
            return dict(point_masses=self.point_masses(time))
        return dict()

    def primitive_array(self, t, mesh, i0=0, i1=None, num_fields=None):
        """Fallback: the reference's per-point loop (exact by construction)."""
        i1 = mesh.ni if i1 is None else i1
        out = np.zeros([i1 - i0, mesh.nj, num_fields])
        for i in range(i0, i1):
            for j in range(mesh.nj):
                self.primitive(t, mesh.cell_coordinates(i, j), out[i - i0, j])
        return out

    def primitive_points(self, t, x, y, num_fields):
        """Primitive data at arbitrary points (arrays x, y of one shape); fallback: per-point loop."""
        out = np.zeros(x.shape + (num_fields,))
        for idx in np.ndindex(*x.shape):
            self.primitive(t, (float(x[idx]), float(y[idx])), out[idx])
        return out


class CircumbinaryDisk(Setup):
    eos = "isothermal"
    domain_radius = 12.0
    mach_number = 10.0
    eccentricity = 0.0
    mass_ratio = 1.0
    sink_rate = 10.0
    sink_radius = 0.05
    softening_length = 0.05
    buffer_is_enabled = True
    sink_model = "torque_free"
    initial_sigma = 1.0
    initial_pressure = 1e-2
    cooling_coefficient = 0.0
    alpha = 0.1
    nu = 0.001
    constant_softening = True
    gamma_law_index = 5.0 / 3.0
    which_diagnostics = "none"

    reference_time_scale = 2.0 * pi
    default_end_time = 1000.0
    default_resolution = 512

    def validate(self):
        if self.eos not in ("isothermal", "gamma-law"):
            raise SetupError(f"eos must be isothermal or gamma-law, got {self.eos}")
        if self.which_diagnostics not in ("none", "mdots"):
            raise SetupError(f"which_diagnostics must be none or mdots, got {self.which_diagnostics}")

    @property
    def is_isothermal(self):
        return self.eos == "isothermal"

    @property
    def is_gamma_law(self):
        return self.eos == "gamma-law"

    @property
    def num_fields(self):
        return 3 if self.is_isothermal else 4

    def primitive(self, t, coords, primitive):
        """Per-point initial data, circumbinary_disk.py:82-108."""
        x, y = coords
        r = sqrt(x * x + y * y)
        rs = sqrt(x * x + y * y + self.softening_length * self.softening_length)
        phx = -y / max(r, 1e-12)
        phy = +x / max(r, 1e-12)
        if self.is_isothermal:
            primitive[0] = self.initial_sigma
            primitive[1] = sqrt(1.0 / rs) * phx
            primitive[2] = sqrt(1.0 / rs) * phy
        else:
            cavity = 0.0001 + 0.9999 * exp(-((1.0 / rs) ** 30))
            primitive[0] = self.initial_sigma * rs ** (-3.0 / 5.0) * cavity
            primitive[1] = sqrt(1.0 / rs) * phx
            primitive[2] = sqrt(1.0 / rs) * phy
            primitive[3] = self.initial_pressure * rs ** (-3.0 / 2.0) * cavity

    def primitive_array(self, t, mesh, i0=0, i1=None, num_fields=None):
        """
        Rows [i0, i1) of the initial data without the interpreter loop.  The isothermal branch uses
        only + * / sqrt max, all correctly rounded in numpy, so it is bit-identical
        to the per-point loop (checked in tests).  The gamma-law branch calls
        pow/exp, where numpy's vector kernels differ from libm in the last
        place at ~5 % of the zones: those rows come from libsfhip.so's host routine
        (the per-point formulas with the interpreter's own libm calls, threaded:
        bit-identical, tested at 8192^2); without the library small meshes go through
        the per-point loop and large ones through numpy.
        """
        i1 = mesh.ni if i1 is None else i1
        if self.is_gamma_law:
            exact = self._gamma_law_rows_native(mesh, i0, i1)
            if exact is not None:
                return exact
            if (i1 - i0) * mesh.nj <= (1 << 18):
                return Setup.primitive_array(self, t, mesh, i0, i1, 4)
        i = np.arange(i0, i1, dtype=np.float64)[:, None]
        j = np.arange(mesh.nj, dtype=np.float64)[None, :]
        x = mesh.x0 + (i + 0.5) * mesh.dx
        y = mesh.y0 + (j + 0.5) * mesh.dy
        return self._primitive_vectorised(x, y, (i1 - i0, mesh.nj))

    def _gamma_law_rows_native(self, mesh, i0, i1):
        """Rows [i0, i1) of the gamma-law disk from libsfhip.so's HOST routine (sfhip_host_gamma_law_disk_primitive:
        the per-point formulas with the libm calls the interpreter makes, threaded; bit-identical to ``primitive``),
        or None where the library is not available (CPU-only checks of the host logic: the slower paths below)."""
        import ctypes

        try:
            from .kernel.library import Library

            fn = Library(mode="hip").module.sfhip_host_gamma_law_disk_primitive
        except Exception:
            return None
        d, i = ctypes.c_double, ctypes.c_int
        fn.argtypes = [i, i, d, d, d, d, i, d, d, d, d, d, i, ctypes.POINTER(d)]
        fn.restype = i
        out = np.empty([i1 - i0, mesh.nj, 4])
        status = fn(i1 - i0, mesh.nj, mesh.x0, mesh.y0, mesh.dx, mesh.dy, i0,
                    self.softening_length * self.softening_length, float(self.initial_sigma), float(self.initial_pressure),
                    -3.0 / 5.0, -3.0 / 2.0, 0, out.ctypes.data_as(ctypes.POINTER(d)))
        if status != 0:
            raise SetupError("sfhip_host_gamma_law_disk_primitive failed")
        return out

    def primitive_points(self, t, x, y, num_fields=None):
        if self.is_gamma_law and x.size <= (1 << 18):
            return Setup.primitive_points(self, t, x, y, 4)
        return self._primitive_vectorised(x, y, x.shape)

    @property
    def has_device_primitive(self):
        """The isothermal disk's initial data can be made on the device bit for bit (only + * / sqrt max); the
        gamma-law disk's pow/exp cannot (glibc's are not correctly rounded and no device library repeats their
        last-place choices), so that branch is made by libsfhip.so's threaded host routine and copied."""
        return self.is_isothermal

    def device_primitive(self, lib, t, mesh, i0, i1, out):
        """
        Rows [i0, i1) of the initial data written straight into the interior of the device array `out`
        ((i1 - i0 + 4, nj + 4, 3), guards untouched) by sfhip_circumbinary_disk_primitive: no host array, no
        copy (the reference: a Python loop over every zone, cbdiso_2d.py:33-46).  Bit-identical to ``primitive``.
        """
        if not self.has_device_primitive:
            raise SetupError("initial data of the gamma-law disk are made on the host")
        lib.sfhip_circumbinary_disk_primitive[(i1 - i0, mesh.nj)](
            mesh.x0, mesh.y0, mesh.dx, mesh.dy, i0, self.softening_length * self.softening_length,
            float(self.initial_sigma), out)

    def _primitive_vectorised(self, x, y, shape):
        r2 = x * x + y * y
        r = np.sqrt(r2)
        rs = np.sqrt(r2 + self.softening_length * self.softening_length)
        rc = np.maximum(r, 1e-12)
        vk = np.sqrt(1.0 / rs)
        out = np.zeros([*shape, self.num_fields])
        out[..., 1] = vk * (-y / rc)
        out[..., 2] = vk * (+x / rc)
        if self.is_isothermal:
            out[..., 0] = self.initial_sigma
        else:
            cavity = 0.0001 + 0.9999 * np.exp(-((1.0 / rs) ** 30))
            out[..., 0] = self.initial_sigma * rs ** (-3.0 / 5.0) * cavity
            out[..., 3] = self.initial_pressure * rs ** (-3.0 / 2.0) * cavity
        return out

    def mesh(self, resolution):
        return PlanarCartesian2DMesh.centered_square(self.domain_radius, resolution)

    @property
    def physics(self):
        """circumbinary_disk.py:117-153"""
        if self.is_isothermal:
            return dict(
                eos_type=EquationOfState.LOCALLY_ISOTHERMAL,
                mach_number=self.mach_number,
                point_mass_function=self.point_masses,
                buffer_is_enabled=self.buffer_is_enabled,
                buffer_driving_rate=100.0,
                buffer_onset_width=1.0,
                cooling_coefficient=0.0,
                constant_softening=self.constant_softening,
                viscosity_model=ViscosityModel.CONSTANT_NU if self.nu > 0.0 else ViscosityModel.NONE,
                viscosity_coefficient=self.nu,
                alpha=0.0,
                diagnostics=self.diagnostics,
            )
        return dict(
            eos_type=EquationOfState.GAMMA_LAW,
            gamma_law_index=self.gamma_law_index,
            point_mass_function=self.point_masses,
            buffer_is_enabled=self.buffer_is_enabled,
            buffer_driving_rate=1000.0,
            buffer_onset_width=0.1,
            cooling_coefficient=self.cooling_coefficient,
            constant_softening=self.constant_softening,
            viscosity_model=ViscosityModel.CONSTANT_ALPHA if self.alpha > 0.0 else ViscosityModel.NONE,
            viscosity_coefficient=0.0,
            alpha=self.alpha,
            diagnostics=self.diagnostics,
        )

    @property
    def diagnostics(self):
        if self.which_diagnostics == "none":
            return []
        return [
            dict(quantity="time"),
            dict(quantity="mdot", which_mass=1, accretion=True),
            dict(quantity="mdot", which_mass=2, accretion=True),
        ]

    @property
    def solver(self):
        return "cbdiso_2d" if self.is_isothermal else "cbdgam_2d"

    @property
    def orbital_elements(self):
        return OrbitalElements(
            semimajor_axis=1.0, total_mass=1.0, mass_ratio=self.mass_ratio, eccentricity=self.eccentricity
        )

    def kepler_binary(self):
        """
        What `point_masses(time)` is made of, for callers that evaluate it without Python (sfhip_plan_step_rk2):
        the four orbital elements and the four constants both masses share (sink model as its integer value).
        """
        el = self.orbital_elements
        return ((float(el.semimajor_axis), float(el.total_mass), float(el.mass_ratio), float(el.eccentricity)),
                (float(self.softening_length), float(self.sink_rate), float(self.sink_radius),
                 float(SinkModel[self.sink_model.upper()].value)))

    def point_masses(self, time):
        """circumbinary_disk.py:194-212"""
        common = dict(
            softening_length=self.softening_length,
            sink_model=SinkModel[self.sink_model.upper()],
            sink_rate=self.sink_rate,
            sink_radius=self.sink_radius,
        )
        return tuple(PointMass(**common, **body._asdict()) for body in self.orbital_elements.orbital_state(time))


class KitpCodeComparison(Setup):
    """
    The code-comparison disk of the reference (``sailfish/setups/circumbinary_disk.py:218-375``): a
    locally isothermal disk with a depleted cavity inside r = 2.5, sub-Keplerian rotation blended into
    the binary's frequency near the centre, and a small radial kick.  It is the circumbinary setup that
    can select the discontinuous-Galerkin solver (``use_dg``).
    """

    mach_number = 10.0
    eccentricity = 0.0
    mass_ratio = 1.0
    sink_radius = 0.05
    softening_length = 0.05
    nu = 0.001
    single_point_mass = False
    sink_model = "acceleration_free"
    domain_radius = 8.0
    sink_rate = 10.0
    buffer_is_enabled = True
    use_dg = False
    disk_kick = 1e-4
    which_diagnostics = "kitp"

    reference_time_scale = 2.0 * pi
    default_end_time = 1.0
    default_resolution = 200

    def validate(self):
        if self.which_diagnostics not in ("kitp", "forces"):
            raise SetupError("Unknown option for diagnostics.")

    def primitive(self, t, coords, primitive):
        """Per-point initial data, circumbinary_disk.py:242-263 (Python floats, as there)."""
        x, y = coords
        r = sqrt(x * x + y * y)
        sigma = 1.0 * (1e-5 + (1 - 1e-5) * exp(-((2.5 / r) ** 12)))
        omega_b = (1.0 / 1.0**3) ** 0.5
        omega_0 = (1.0 / r**3 * (1.0 - 1.0 / self.mach_number**2)) ** 0.5
        omega = (omega_0**-4.0 + omega_b**-4.0) ** (-1 / 4.0)
        kick = self.disk_kick * y * exp(-((r / 3.5) ** 6))
        primitive[0] = sigma
        primitive[1] = omega * -y + kick * x / r
        primitive[2] = omega * +x + kick * y / r

    def primitive_array(self, t, mesh, i0=0, i1=None, num_fields=None):
        i1 = mesh.ni if i1 is None else i1
        if (i1 - i0) * mesh.nj <= (1 << 18):
            return Setup.primitive_array(self, t, mesh, i0, i1, 3)
        i = np.arange(i0, i1, dtype=np.float64)[:, None]
        j = np.arange(mesh.nj, dtype=np.float64)[None, :]
        x = np.broadcast_to(mesh.x0 + (i + 0.5) * mesh.dx, (i1 - i0, mesh.nj))
        y = np.broadcast_to(mesh.y0 + (j + 0.5) * mesh.dy, (i1 - i0, mesh.nj))
        return self._primitive_vectorised(x, y)

    def primitive_points(self, t, x, y, num_fields=None):
        """Small inputs take the exact per-point loop; large ones numpy (pow/exp: last-place differences)."""
        if x.size <= (1 << 18):
            return Setup.primitive_points(self, t, x, y, 3)
        return self._primitive_vectorised(x, y)

    def _primitive_vectorised(self, x, y):
        r = np.sqrt(x * x + y * y)
        sigma = 1.0 * (1e-5 + (1 - 1e-5) * np.exp(-((2.5 / r) ** 12)))
        omega_0 = (1.0 / r**3 * (1.0 - 1.0 / self.mach_number**2)) ** 0.5
        omega = (omega_0**-4.0 + 1.0) ** (-1 / 4.0)
        kick = self.disk_kick * y * np.exp(-((r / 3.5) ** 6))
        out = np.empty(x.shape + (3,))
        out[..., 0] = sigma
        out[..., 1] = omega * -y + kick * x / r
        out[..., 2] = omega * +x + kick * y / r
        return out

    def mesh(self, resolution):
        return PlanarCartesian2DMesh.centered_square(self.domain_radius, resolution)

    @property
    def diagnostics(self):
        """circumbinary_disk.py:272-302"""
        head = [
            dict(quantity="time"),
            dict(quantity="mdot", which_mass=1, accretion=True),
            dict(quantity="mdot", which_mass=2, accretion=True),
        ]
        if self.which_diagnostics == "kitp":
            return head + [
                dict(quantity="torque", which_mass="both", gravity=True),
                dict(quantity="torque", which_mass="both", gravity=True, radial_cut=(1.0, self.domain_radius)),
                dict(quantity="sigma_m1"),
                dict(quantity="eccentricity_vector", radial_cut=(1.0, 6.0)),
            ]
        return head + [
            dict(quantity=q, which_mass=m, **{term: True})
            for m in (1, 2) for q in ("fx", "fy") for term in ("gravity", "accretion")
        ]

    @property
    def physics(self):
        """circumbinary_disk.py:304-318"""
        return dict(
            eos_type=EquationOfState.LOCALLY_ISOTHERMAL,
            mach_number=self.mach_number,
            point_mass_function=self.point_masses,
            buffer_is_enabled=self.buffer_is_enabled,
            buffer_driving_rate=100.0,
            buffer_onset_width=1.0,
            viscosity_coefficient=self.nu,
            viscosity_model=ViscosityModel.CONSTANT_NU if self.nu > 0.0 else ViscosityModel.NONE,
            diagnostics=self.diagnostics,
        )

    @property
    def solver(self):
        return "cbdisodg_2d" if self.use_dg else "cbdiso_2d"  # circumbinary_disk.py:320-322

    @property
    def orbital_elements(self):
        return OrbitalElements(
            semimajor_axis=1.0, total_mass=1.0, mass_ratio=self.mass_ratio, eccentricity=self.eccentricity
        )

    def point_masses(self, time):
        """circumbinary_disk.py:345-372"""
        common = dict(
            softening_length=self.softening_length,
            sink_model=SinkModel[self.sink_model.upper()],
            sink_rate=self.sink_rate,
            sink_radius=self.sink_radius,
        )
        if self.single_point_mass:
            return PointMass(mass=1.0, **common)
        return tuple(PointMass(**common, **body._asdict()) for body in self.orbital_elements.orbital_state(time))


class CylindricalExplosion(Setup):
    eos = "isothermal"
    smooth = 6.0
    use_dg = False

    default_end_time = 0.3
    default_resolution = 200

    def validate(self):
        if self.eos not in ("isothermal", "gamma-law"):
            raise ValueError(f"eos must be isothermal or gamma-law, got {self.eos}")
        if self.use_dg and not self.is_isothermal:
            raise ValueError("DG mode is only available for eos=isothermal")  # simple2d.py:113-114

    @property
    def is_isothermal(self):
        return self.eos == "isothermal"

    @property
    def num_fields(self):
        return 3 if self.is_isothermal else 4

    def primitive(self, t, coords, primitive):
        """simple2d.py:65-79"""
        x, y = coords
        r = (x * x + y * y) ** 0.5
        f = exp(-((r / 0.25) ** self.smooth)) if self.smooth != 0.0 else float(r < 0.25)
        if self.is_isothermal:
            primitive[0] = 0.1 + 0.9 * f
        else:
            primitive[0] = 0.100 + 0.900 * f
            primitive[3] = 0.125 + 0.875 * f

    def primitive_array(self, t, mesh, i0=0, i1=None, num_fields=None):
        return Setup.primitive_array(self, t, mesh, i0, i1, self.num_fields)

    def mesh(self, resolution):
        return PlanarCartesian2DMesh.centered_square(1.0, resolution)

    @property
    def physics(self):
        if self.is_isothermal:
            return dict(eos_type=EquationOfState.GLOBALLY_ISOTHERMAL, sound_speed=1.0)
        return dict(eos_type=EquationOfState.GAMMA_LAW, gamma_law_index=5 / 3)

    @property
    def solver(self):
        if self.is_isothermal:
            return "cbdisodg_2d" if self.use_dg else "cbdiso_2d"  # simple2d.py:94
        return "cbdgam_2d"


class UniformPolar(Setup):
    """Uniform gas at rest on a polar mesh: exercises srhd_2d's geometrical source terms (simple2d.py:13-37)."""

    default_end_time = 1.0
    default_resolution = 100

    def primitive(self, t, coords, primitive):
        primitive[0] = 1.0
        primitive[1] = 0.0
        primitive[2] = 0.0
        primitive[3] = 1.0

    def mesh(self, num_zones_per_decade):
        return LogSphericalMesh(1.0, 50.0, num_zones_per_decade, polar_grid=True)

    @property
    def physics(self):
        return dict()

    @property
    def solver(self):
        return "srhd_2d"


class _WindEnvelope:
    """
    The ballistic medium of the envelope-shock problem (envelope_shock.py:12-88): shells of mass
    coordinate m leave the origin at time m / mdot_wind with gamma-beta u_wind + (m / m1)^-psi and
    coast.  Expressions are kept operation for operation: the initial data must carry the
    reference's bits.
    """

    t_delay = 1.0
    m1 = 1.0
    psi = 0.25
    m_cloud = 1e5 * m1
    u_wind = 0.1
    mdot_wind = m_cloud / t_delay

    @classmethod
    def launch_time(cls, m):
        return m / cls.mdot_wind

    @classmethod
    def gamma_beta(cls, m):
        return cls.u_wind + (m / cls.m1) ** (-cls.psi)

    @classmethod
    def gamma_beta_prime(cls, m):
        return -((m / cls.m1) ** (-cls.psi)) * cls.psi / m

    @classmethod
    def speed(cls, m):
        u = cls.gamma_beta(m)
        return u / (1 + u**2) ** 0.5

    @classmethod
    def speed_prime(cls, m):
        u = cls.gamma_beta(m)
        du_dm = cls.gamma_beta_prime(m)
        dv_du = (1 + u**2) ** (-3 / 2)
        return dv_du * du_dm

    @classmethod
    def radius(cls, m, t):
        return cls.speed(m) * (t - cls.launch_time(m))

    @classmethod
    def density(cls, m, t):
        t0 = cls.launch_time(m)
        t0_prime = 1.0 / cls.mdot_wind
        u = cls.gamma_beta(m)
        u_prime = cls.gamma_beta_prime(m)
        mdot_inverse = t0_prime - u_prime / u * (t - t0) / (1 + u**2)
        r = cls.radius(m, t)
        return 1.0 / (4 * pi * r**2 * u * mdot_inverse)

    @classmethod
    def mass_at(cls, r, t):
        """Newton iteration for the shell at radius r (envelope_shock.py:64-88)."""
        m = 1e-12
        n = 0
        while True:
            fm = r - cls.radius(m, t)
            v = cls.speed(m)
            t0 = cls.launch_time(m)
            dv = cls.speed_prime(m)
            dt0 = 1.0 / cls.mdot_wind
            gm = -(dv * (t - t0) - v * dt0)
            m -= fm / gm
            if abs(fm) < 1e-10:
                return m
            if n > 200:
                raise ValueError("too many iterations")
            n += 1


class EnvelopeShock(Setup):
    """
    A relativistic shell or jet launched into a homologous relativistic envelope
    (envelope_shock.py:91-264); with ``polar_extent > 0`` it is the srhd_2d workload of
    BASELINE.json configs[4].  The 1-d variant (srhd_1d) is not built here.
    """

    u_shell = 30.0
    m_shell = 1.0
    w_shell = 1.0
    q_shell = 0.1
    t_start = 1.0
    r_inner = 0.1
    r_outer = 1.0
    expand = True
    jet_energy = 0.0
    jet_gamma_beta = 10.0
    jet_theta = 0.1
    jet_duration = 1.0
    polar_extent = 0.5

    default_end_time = 1.0
    default_resolution = 800

    def validate(self):
        if self.polar_extent <= 0.0:
            raise SetupError("this package builds the 2d (srhd_2d) variant only: polar_extent must be > 0")
        if self.jet_energy > 0.0 and self.u_shell > 0.0:
            raise SetupError("we don't simulate a jet and a shell at the same time")

    @property
    def start_time(self):
        return self.t_start

    def envelope_state(self, r, t):
        m = _WindEnvelope.mass_at(r, t)
        d = _WindEnvelope.density(m, t)
        u = _WindEnvelope.gamma_beta(m)
        return m, d, u, 1e-6 * d

    def shell_u_profile_mass(self, m):
        if m < self.m_shell:
            return 0.0
        return self.u_shell * exp(-(m / self.m_shell - 1.0) / self.w_shell)

    def shell_u_profile_polar(self, q):
        return exp(-((q / self.q_shell) ** 2))

    def primitive(self, t, coord, primitive):
        r, q = coord
        m, d, u, p = self.envelope_state(r, t)
        primitive[0] = d
        primitive[1] = u + self.shell_u_profile_mass(m) * self.shell_u_profile_polar(q)
        primitive[2] = 0.0
        primitive[3] = p

    def primitive_rows(self, t, mesh, i0, i1):
        """Rows [i0, i1) at once: the radial profile is evaluated once per row (the reference caches it per
        (r, t), envelope_shock.py:243), the polar profile once per column; same scalar calls, same bits."""
        nj = mesh.shape[1]
        out = np.zeros([i1 - i0, nj, 4])
        polar = np.array([self.shell_u_profile_polar(mesh.cell_coordinates(t, 0, j)[1]) for j in range(nj)])
        for i in range(i0, i1):
            m, d, u, p = self.envelope_state(mesh.cell_coordinates(t, i, 0)[0], t)
            du = self.shell_u_profile_mass(m)
            out[i - i0, :, 0] = d
            out[i - i0, :, 1] = u + du * polar  # elementwise: the same multiply and add per zone
            out[i - i0, :, 3] = p
        return out

    @property
    def physics(self):
        if self.jet_energy == 0.0:
            return dict()
        jet_gamma = (self.jet_gamma_beta * self.jet_gamma_beta + 1.0) ** 0.5
        jet_mdot = self.jet_energy / ((jet_gamma - 1.0) * self.jet_duration / _WindEnvelope.m_cloud)
        return dict(jet_mdot=jet_mdot, jet_gamma_beta=self.jet_gamma_beta, jet_theta=self.jet_theta,
                    jet_duration=self.jet_duration)

    def mesh(self, num_zones_per_decade):
        return LogSphericalMesh(
            r0=self.r_inner, r1=self.r_outer, num_zones_per_decade=num_zones_per_decade,
            scale_factor_derivative=(1.0 / self.t_start) if self.expand else None,
            polar_grid=True, polar_extent=self.polar_extent * pi,
        )

    @property
    def solver(self):
        return "srhd_2d"

    @property
    def boundary_condition(self):
        return ("jet", "outflow") if self.jet_energy > 0.0 else "outflow"


SETUPS = {
    "circumbinary-disk": CircumbinaryDisk, "kitp-code-comparison": KitpCodeComparison,
    "cylindrical-explosion": CylindricalExplosion,
    "uniform-polar": UniformPolar, "envelope-shock": EnvelopeShock,
}


def make_setup(name, **model_parameters):
    try:
        return SETUPS[name](**model_parameters)
    except KeyError:
        raise SetupError(f"no setup named {name}; have {sorted(SETUPS)}")
