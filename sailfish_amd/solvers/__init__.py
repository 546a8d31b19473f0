"""
Solver registry (reference: ``sailfish/solvers/__init__.py:13-46``).

``make_solver(name, physics, options, **kwargs)`` constructs the named solver;
``register_solver_extension`` is the reference's own zero-patch injection point,
so a ``.sailfish`` config listing ``sailfish_amd.solvers.cbdiso_2d`` as a solver
extension routes the reference driver to this implementation.
"""


class SolverInitializationError(Exception):
    """An invalid runtime configuration"""


extension_solvers = dict()


def register_solver_extension(solver_name, solver_module):
    extension_solvers[solver_name] = solver_module


def make_solver(name, physics, options, **kwargs):
    from . import cbdgam_2d, cbdiso_2d, cbdisodg_2d, srhd_2d

    solvers = dict(cbdiso_2d=cbdiso_2d, cbdgam_2d=cbdgam_2d, srhd_2d=srhd_2d, cbdisodg_2d=cbdisodg_2d,
                   **extension_solvers)
    if name not in solvers:
        raise SolverInitializationError(f"no solver named {name}; this package builds {sorted(solvers)}")
    try:
        return solvers[name].Solver(physics=dict(physics or {}), options=dict(options or {}), **kwargs)
    except (TypeError, ValueError) as e:
        raise SolverInitializationError(e)
