"""
The one mesh the cbdiso_2d / cbdgam_2d solvers accept: a uniform 2-d Cartesian
grid (reference: ``sailfish/mesh.py:175-261``).  Patch extents handed to the
kernels come from ``vertex_coordinates`` and must be computed with exactly these
expressions for multi-patch runs to reproduce the reference's coordinates.
"""
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
