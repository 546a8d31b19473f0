"""
Meshes of the solvers built here.

``PlanarCartesian2DMesh`` is the one mesh the cbdiso_2d / cbdgam_2d solvers accept: a
uniform 2-d Cartesian grid (reference: ``sailfish/mesh.py:175-261``).  Patch extents
handed to the kernels come from ``vertex_coordinates`` and must be computed with exactly
these expressions for multi-patch runs to reproduce the reference's coordinates.

``LogSphericalMesh`` (reference: ``sailfish/mesh.py:52-172``) is srhd_2d's mesh:
logarithmic radial zones, uniform polar zones, optional homologous expansion.
"""
from math import log, log10, pi
from typing import NamedTuple


class PlanarCartesian2DMesh(NamedTuple):
    x0: float = 0.0
    y0: float = 0.0
    x1: float = 1.0
    y1: float = 1.0
    ni: int = 1000
    nj: int = 1000

    def __str__(self):
        return (
            f"<planar cartesian 2d: ({self.x0} -> {self.x1}) x "
            f"({self.y0} -> {self.y1}), shape {self.shape}>"
        )

    @classmethod
    def centered_square(cls, domain_radius, resolution):
        r = domain_radius
        return cls(-r, -r, +r, +r, resolution, resolution)

    @classmethod
    def centered_rectangle(cls, height, resolution, aspect):
        if type(aspect) is not int:
            raise ValueError("centered_rectangle requires aspect to be int")
        return cls(
            -0.5 * height * aspect, -0.5 * height, +0.5 * height * aspect, +0.5 * height,
            resolution * aspect, resolution,
        )

    @property
    def dx(self):
        return (self.x1 - self.x0) / self.ni

    @property
    def dy(self):
        return (self.y1 - self.y0) / self.nj

    @property
    def shape(self):
        return self.ni, self.nj

    @property
    def num_total_zones(self):
        return self.ni * self.nj

    def min_spacing(self, time=None):
        return min(self.dx, self.dy)

    def cell_coordinates(self, i, j):
        return self.x0 + (i + 0.5) * self.dx, self.y0 + (j + 0.5) * self.dy

    def vertex_coordinates(self, i, j):
        """Lower-left corner of zone (i, j)."""
        return self.x0 + i * self.dx, self.y0 + j * self.dy

    def sub_mesh(self, di, dj):
        xa, ya = self.vertex_coordinates(di[0], dj[0])
        xb, yb = self.vertex_coordinates(di[1], dj[1])
        return PlanarCartesian2DMesh(xa, ya, xb, yb, di[1] - di[0], dj[1] - dj[0])


class LogSphericalMesh(NamedTuple):
    """
    Logarithmically spaced radial zones (and, with ``polar_grid``, uniformly spaced polar
    zones); comoving coordinates are fixed, proper distances scale with
    ``scale_factor(time) = scale_factor_derivative * time`` when the mesh expands.
    Face positions, zone counts and the polar spacing use the reference's expressions
    (mesh.py:60-172) so that the kernels receive identical numbers.
    """

    r0: float = 1.0
    r1: float = 10.0
    num_zones_per_decade: int = 1000
    scale_factor_derivative: float = None
    polar_grid: bool = False
    polar_extent: float = pi

    def __str__(self):
        if self.scale_factor_derivative is None:
            motion = "no expansion"
        else:
            motion = f"homologous with a-dot = {self.scale_factor_derivative:0.2f}"
        return (f"<log spherical: ({self.r0} -> {self.r1}), {motion}, "
                f"{self.num_zones_per_decade} zones per decade, shape {self.shape}>")

    def scale_factor(self, time):
        return 1.0 if self.scale_factor_derivative is None else self.scale_factor_derivative * time

    def min_spacing(self, time=None):
        a, b = self.faces(0, 1)
        return (b - a) * self.scale_factor(time)

    @property
    def num_radial_zones(self):
        return int(log10(self.r1 / self.r0) * self.num_zones_per_decade)

    @property
    def num_polar_zones(self):
        if not self.polar_grid:
            raise ValueError("only defined for a 2D spherical polar mesh")
        return int(self.num_zones_per_decade * self.polar_extent / log(10))

    @property
    def polar_spacing(self):
        return self.polar_extent / self.num_polar_zones

    @property
    def shape(self):
        return (self.num_radial_zones, self.num_polar_zones) if self.polar_grid else (self.num_radial_zones,)

    @property
    def num_total_zones(self):
        total = 1
        for n in self.shape:
            total *= n
        return total

    def faces(self, i0=0, i1=None):
        """Comoving radial face positions of zones [i0, i1): i1 - i0 + 1 values."""
        if i1 is None:
            i1 = self.shape[0]
        k = 1.0 / self.num_zones_per_decade
        return [self.r0 * 10 ** (i * k) for i in range(i0, i1 + 1)]

    def zone_center(self, t, i):
        k = 1.0 / self.num_zones_per_decade
        return self.r0 * 10 ** ((i + 0.5) * k) * self.scale_factor(t)

    def zone_centers(self, t, i0=0, i1=None):
        return [self.zone_center(t, i) for i in range(i0, i1 or self.shape[0])]

    def cell_coordinates(self, t, i, j):
        return self.zone_center(t, i), (j + 0.5) * self.polar_spacing

    def radial_vertices(self, time):
        a = self.scale_factor(time)
        return [r * a for r in self.faces()]

    @property
    def polar_vertices(self):
        return [j * self.polar_spacing for j in range(self.num_polar_zones + 1)]
