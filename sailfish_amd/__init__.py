"""
sailfish_amd -- MI355X (gfx950) implementation of sailfish's cbdiso_2d / cbdgam_2d
finite-volume update, behind the reference's own `kernel.Library` / `Solver` API.

Only the hot path is here: the kernel library loader, the two
solvers, slab decomposition with halo exchange, and the driver's dt / Mzps loop.
"""

__version__ = "0.1.0"
