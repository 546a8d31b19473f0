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


def solve_newton_raphson(f, g, x):
    """Newton iteration to |f| <= 1e-15, at most 10 corrections (kepler.py:414-421)."""
    n = 0
    while abs(f(x)) > 1e-15:
        x -= f(x) / g(x)
        n += 1
        if n > 10:
            raise ValueError("solve_newton_rapheson: no solution")
    return x


class OrbitalElements(NamedTuple):
    semimajor_axis: float
    total_mass: float
    mass_ratio: float
    eccentricity: float

    @property
    def omega(self):
        a = self.semimajor_axis
        return sqrt(NEWTON_G * self.total_mass / a / a / a)

    @property
    def period(self):
        return 2.0 * pi / self.omega

    def eccentric_anomaly(self, time_since_periapse):
        p = self.period
        t = time_since_periapse - self.period * floor(time_since_periapse / p)
        e = self.eccentricity
        n = self.omega * t  # mean anomaly
        return solve_newton_raphson(lambda k: k - e * sin(k) - n, lambda k: 1.0 - e * cos(k), n)

    def orbital_state_from_eccentric_anomaly(self, eccentric_anomaly):
        a, m, q, e, w = self.semimajor_axis, self.total_mass, self.mass_ratio, self.eccentricity, self.omega
        m1 = m / (1.0 + q)
        m2 = m - m1
        ck = cos(eccentric_anomaly)
        sk = sin(eccentric_anomaly)
        x1 = -a * q / (1.0 + q) * (e - ck)
        y1 = +a * q / (1.0 + q) * (sk) * sqrt(1.0 - e * e)
        x2 = -x1 / q
        y2 = -y1 / q
        vx1 = -a * q / (1.0 + q) * w / (1.0 - e * ck) * sk
        vy1 = +a * q / (1.0 + q) * w / (1.0 - e * ck) * ck * sqrt(1.0 - e * e)
        vx2 = -vx1 / q
        vy2 = -vy1 / q
        return OrbitalBody(m1, x1, y1, vx1, vy1), OrbitalBody(m2, x2, y2, vx2, vy2)

    def orbital_state(self, time_since_periapse):
        return self.orbital_state_from_eccentric_anomaly(self.eccentric_anomaly(time_since_periapse))
