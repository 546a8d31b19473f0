"""
Two-body orbit -> positions and velocities of the binary components.

Only the forward problem the hot path needs (``OrbitalElements.orbital_state``,
reference ``sailfish/physics/kepler.py:290-373, 414-421``): it runs on the host
once per RK stage per patch and produces 10 of the kernel scalars, so it has to
reproduce the reference's floating-point expressions exactly (pinned by
tests/golden/kepler.npz).
"""
from math import cos, floor, pi, sin, sqrt
from typing import NamedTuple

NEWTON_G = 1.0


class OrbitalBody(NamedTuple):
    mass: float
    position_x: float
    position_y: float
    velocity_x: float
    velocity_y: float


KEPLER_TOLERANCE = 1e-15   # on the residual of Kepler's equation (kepler.py:416)
KEPLER_MAX_CORRECTIONS = 11  # the reference gives up once an 11th correction has been applied (kepler.py:417-420)


class OrbitalElements(NamedTuple):
    semimajor_axis: float
    total_mass: float
    mass_ratio: float
    eccentricity: float

    @property
    def omega(self):
        """Orbital angular frequency sqrt(G M / a^3), divided out one factor of a at a time as the reference does."""
        a = self.semimajor_axis
        return sqrt(NEWTON_G * self.total_mass / a / a / a)

    @property
    def period(self):
        return 2.0 * pi / self.omega

    def eccentric_anomaly(self, time_since_periapse):
        """
        Kepler's equation E - e sin E = M by Newton's method started from E = M (kepler.py:356-366, 414-421).
        The time is first reduced to one orbit; the iteration stops at |residual| <= 1e-15.
        """
        period = self.period
        t = time_since_periapse - self.period * floor(time_since_periapse / period)
        e = self.eccentricity
        mean_anomaly = self.omega * t
        E = mean_anomaly
        for _ in range(KEPLER_MAX_CORRECTIONS):
            residual = E - e * sin(E) - mean_anomaly
            if not abs(residual) > KEPLER_TOLERANCE:
                return E
            E -= residual / (1.0 - e * cos(E))
        raise ValueError("eccentric_anomaly: Newton's method did not converge")

    def orbital_state_from_eccentric_anomaly(self, eccentric_anomaly):
        """
        Positions and velocities of the two bodies about their centre of mass (kepler.py:329-354).  Body 1 sits
        on a lever arm a q / (1 + q), body 2 opposite at 1/q of that; every product and quotient below is taken
        in the reference's left-to-right order (a sign flip is exact, so where the sign enters does not matter),
        because these ten numbers become kernel arguments and must carry the reference's bits
        (tests/golden/kepler.npz).
        """
        a, e, q = self.semimajor_axis, self.eccentricity, self.mass_ratio
        mass1 = self.total_mass / (1.0 + q)
        mass2 = self.total_mass - mass1
        cosE, sinE = cos(eccentric_anomaly), sin(eccentric_anomaly)
        arm = a * q / (1.0 + q)
        axis_ratio = sqrt(1.0 - e * e)
        rate = arm * self.omega / (1.0 - e * cosE)
        x1 = -arm * (e - cosE)
        y1 = arm * sinE * axis_ratio
        vx1 = -rate * sinE
        vy1 = rate * cosE * axis_ratio
        body1 = OrbitalBody(mass1, x1, y1, vx1, vy1)
        body2 = OrbitalBody(mass2, -x1 / q, -y1 / q, -vx1 / q, -vy1 / q)
        return body1, body2

    def orbital_state(self, time_since_periapse):
        return self.orbital_state_from_eccentric_anomaly(self.eccentric_anomaly(time_since_periapse))
