#!/usr/bin/env python3
"""
Generates the golden fixtures in this directory by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference); the fixtures it writes
are data (inputs/expected outputs as .npz), committed so that the tests can run
where the reference does not exist (the GPU box).

What runs here is the reference itself:
  * its Python classes, imported from /root/reference (setups, Kepler solver,
    Physics, mesh, subdivide, Solver/Patch of cbdiso_2d and cbdgam_2d), and
  * its C kernels, compiled in place from its own sources by
    oracle/build_ref.py (plain gcc on KERNEL_LIB_HEADER + <solver>.c).
The reference builds those kernels through cffi, which is only a build driver
and is not installed in this image; `RefLibrary` below overrides the one method
that would invoke it (Library.load_cpu_module, library.py:187-235) to load the
gcc-built shared object instead.  Everything else -- parse_api, Kernel,
KernelInvocation, to_ctypes, the solvers, set_bc, the RK schedule -- is the
reference's code, unmodified.

Fixtures:
  dg_disk_32.npz               cbdisodg_2d: kitp-code-comparison with use_dg, 32^2, 10 RK2 steps (limiter on, viscosity,
                               buffer, sinks resolved by the coarse mesh)
  dg_disk_48_p3_rk3.npz        same solver, eccentric unequal binary, rk_order 3, 3 patches
  dg_explosion_40.npz          cylindrical-explosion with use_dg, 40^2, 12 steps (the limiter's troubled zones)
  dg_explosion_40_nolimit.npz  same with limit_slopes off, 2 patches
                        dg fixtures store the weights (initial, final), dt[], time[], model and options (JSON).
  kepler.npz            point masses at 50 times for (e, q) in {0, 0.3} x {1, 0.2}
  iso_disk_64.npz       circumbinary-disk, isothermal, 64^2, 20 RK2 steps, 1 patch
  iso_disk_128_p4.npz   same at 128^2 on 4 patches (== 1 patch bitwise)
  iso_disk_rk3_48.npz   rk_order 3, 48^2, 3 patches (48/3 even split), 8 steps
  iso_explosion_64.npz  cylindrical-explosion 64^2, 20 steps
  gam_disk_64.npz       gamma-law disk 64^2, 20 steps (alpha viscosity, buffer)
  gam_cooling_48.npz    gamma-law with cooling 1e-2, 48^2, 10 steps
  gam_explosion_64.npz  gamma-law explosion 64^2, 20 steps
  iso_disk_96_developed.npz, gam_disk_96_developed.npz
                        20 steps of a DEVELOPED disk at 96^2: the reference runs 1500 steps, a second reference solver
                        is started from that solution and time (as from a checkpoint) and recorded
  srhd_uniform_polar.npz        uniform-polar, 20 zones per decade, 10 RK2 steps (geometrical source terms)
  srhd_envelope_shock_p3.npz    envelope-shock (u_shell 30), expanding polar mesh, 3 patches, wavespeed-based dt
  srhd_envelope_jet_p2.npz      envelope-shock with the jet inflow boundary, 2 patches (first-order zones)
  srhd_envelope_rk3_theta2.npz  rk_order 3, plm_theta 2.0, mach_ceiling 100 (the macro-configured build)
                        srhd fixtures store the conserved solution (initial, final), the final primitive,
                        dt[], time[].
Each cbd solver fixture stores: initial primitive, final primitive, dt[], time[].
  iso_reductions_64_p2.npz, gam_reductions_48.npz
                        Solver.reductions() of the reference after 12 steps, for a list of
                        diagnostics covering every quantity / mass / term / radial-cut branch:
                        the state (solution, time), the list (JSON) and the values (complex128).
"""
import os
import sys
from ctypes import CDLL

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, REFERENCE)
sys.path.insert(0, ROOT)


def reference_modules():
    import subprocess

    subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "build_ref.py")], check=True)

    import sailfish.kernel.library as reflib
    import sailfish.setups  # noqa: F401  (registers setup classes)
    import sailfish.solvers.cbdgam_2d as ref_gam
    import sailfish.solvers.cbdiso_2d as ref_iso
    import sailfish.solvers.cbdisodg_2d as ref_dg
    import sailfish.solvers.srhd_2d as ref_srhd
    from sailfish.setup_base import SetupBase
    from sailfish.solvers import make_solver

    class RefLibrary(reflib.Library):
        def load_cpu_module(self, code, name, mode="cpu", define_macros=dict()):
            import numpy

            solver = [n for n in ("cbdiso_2d", "cbdgam_2d", "srhd_2d", "cbdisodg_2d") if f"MODULE: {n}" in code][0]
            suffix = ""
            if solver == "srhd_2d":
                # the macro configurations oracle/build_ref.py compiles (define_macros, srhd_2d.py:225-233)
                macros = (float(define_macros["PLM_THETA"]), float(define_macros["MACH_CEILING"]))
                suffix = {(1.5, 1e6): "", (2.0, 100.0): "_theta2"}[macros]
            self.module = CDLL(os.path.join(ROOT, "oracle", "_ref", f"{solver}_{mode}{suffix}.so"))
            self.xp = numpy

    ref_iso.Library = RefLibrary
    ref_gam.Library = RefLibrary
    ref_srhd.Library = RefLibrary
    ref_dg.Library = RefLibrary
    return SetupBase, make_solver


def run_reference(SetupBase, make_solver, setup_name, resolution, steps, model=None, options=None,
                  num_patches=1, cfl=None, spinup=0):
    """`spinup` > 0: the recorded run starts from a DEVELOPED flow -- the reference runs that many steps first, and
    a second reference solver is then started from the resulting solution and time exactly as the driver restarts
    from a checkpoint (driver.py:439-449); `initial` and `time0` are what that solver was given."""
    setup = SetupBase.find_setup_class(setup_name)(**(model or {}))
    mesh = setup.mesh(resolution)
    solver = make_solver(setup.solver, setup.physics, options or {}, setup=setup, mesh=mesh,
                         time=setup.start_time, solution=None, num_patches=num_patches, mode="cpu")
    cfl = cfl or solver.recommended_cfl
    time0 = None
    if spinup:
        for _ in range(spinup):
            solver.advance(mesh.min_spacing(solver.time) / solver.maximum_wavespeed() * cfl)
        developed, time0 = solver.solution.copy(), solver.time
        assert np.isfinite(developed).all()
        solver = make_solver(setup.solver, setup.physics, options or {}, setup=setup, mesh=mesh,
                             time=time0, solution=developed, num_patches=num_patches, mode="cpu")
    initial = solver.solution.copy()
    dts, times = [], []
    for _ in range(steps):
        # driver.py:508-512
        dx = mesh.min_spacing(solver.time)
        dt = dx / solver.maximum_wavespeed() * cfl
        solver.advance(dt)
        dts.append(dt)
        times.append(solver.time)
    out = dict(initial=initial, final=solver.solution.copy(), dt=np.array(dts), time=np.array(times),
               cfl=np.array(cfl), resolution=np.array(resolution), num_patches=np.array(num_patches))
    if time0 is not None:
        out["time0"] = np.array(time0)
    return out


ISO_DIAGNOSTICS = [
    dict(quantity="time"),
    dict(quantity="mdot", which_mass=1, accretion=True),
    dict(quantity="mdot", which_mass=2, accretion=True),
    dict(quantity="mdot", which_mass="both", accretion=True, radial_cut=(0.0, 0.6)),
    dict(quantity="torque", which_mass="both", gravity=True),
    dict(quantity="torque", which_mass=1, accretion=True),
    dict(quantity="torque", which_mass="both", gravity=True, radial_cut=(1.0, 2.0)),
    dict(quantity="fx", which_mass=1, gravity=True),
    dict(quantity="fy", which_mass=2, gravity=True),
    dict(quantity="power", which_mass=1, gravity=True),
    dict(quantity="power", which_mass=2, accretion=True),
    dict(quantity="sigma_m1"),
    dict(quantity="sigma_m1", radial_cut=(1.0, 2.5)),
    dict(quantity="eccentricity_vector", radial_cut=(1.5, 2.5)),
    dict(quantity="angular_momentum"),
    dict(quantity="mass"),
    dict(quantity="mass", radial_cut=(0.5, 2.0)),
]
GAM_DIAGNOSTICS = [
    dict(quantity="time"),
    dict(quantity="mdot", which_mass=1, accretion=True),
    dict(quantity="mdot", which_mass=2, accretion=True),
    dict(quantity="torque", which_mass="both", gravity=True),
    dict(quantity="torque", which_mass="both", accretion=True, radial_cut=(0.0, 1.0)),
    dict(quantity=3, which_mass="both", gravity=True),
    dict(quantity=3, which_mass=2, accretion=True),
    dict(quantity="sigma_m1"),
    dict(quantity="eccentricity_vector", radial_cut=(1.5, 2.5)),
]


def run_reductions(SetupBase, make_solver, resolution, steps, diagnostics, model=None, num_patches=1):
    import json

    setup = SetupBase.find_setup_class("circumbinary-disk")(**(model or {}))
    mesh = setup.mesh(resolution)
    physics = dict(setup.physics)
    physics["diagnostics"] = diagnostics
    solver = make_solver(setup.solver, physics, {}, setup=setup, mesh=mesh, time=setup.start_time, solution=None,
                         num_patches=num_patches, mode="cpu")
    for _ in range(steps):
        dt = mesh.min_spacing(solver.time) / solver.maximum_wavespeed() * solver.recommended_cfl
        solver.advance(dt)
    values = np.array(solver.reductions(), dtype=np.complex128)
    return dict(solution=solver.solution.copy(), time=np.array(solver.time), values=values,
                diagnostics=np.array(json.dumps(diagnostics)), resolution=np.array(resolution),
                num_patches=np.array(num_patches), model=np.array(json.dumps(model or {})))


def run_srhd(SetupBase, make_solver, setup_name, resolution, steps, model=None, options=None, num_patches=1):
    import json

    setup = SetupBase.find_setup_class(setup_name)(**(model or {}))
    mesh = setup.mesh(resolution)
    solver = make_solver(setup.solver, setup.physics, options or {}, setup=setup, mesh=mesh,
                         time=setup.start_time, solution=None, num_patches=num_patches, mode="cpu")
    initial = solver.solution.copy()
    cfl = solver.recommended_cfl
    dts, times = [], []
    for _ in range(steps):
        dt = mesh.min_spacing(solver.time) / solver.maximum_wavespeed() * cfl  # driver.py:508-510
        solver.advance(dt)
        dts.append(dt)
        times.append(solver.time)
    return dict(initial=initial, final=solver.solution.copy(), final_primitive=solver.primitive.copy(),
                dt=np.array(dts), time=np.array(times), cfl=np.array(cfl), resolution=np.array(resolution),
                num_patches=np.array(num_patches), start_time=np.array(setup.start_time),
                model=np.array(json.dumps(model or {})), options=np.array(json.dumps(options or {})))


def run_dg(SetupBase, make_solver, setup_name, resolution, steps, model=None, options=None, num_patches=1):
    import json

    data = run_reference(SetupBase, make_solver, setup_name, resolution, steps, model, options, num_patches)
    data["model"] = np.array(json.dumps(model or {}))
    data["options"] = np.array(json.dumps(options or {}))
    return data


def kepler_fixture(SetupBase):
    rows = []
    times = np.concatenate([np.linspace(0.0, 13.0, 40), [0.1, 1e-3, 100.7, 6.283185307179586, 31.4, 1e3, 2.5, 7.7, 19.9, 50.0]])
    for e in (0.0, 0.3):
        for q in (1.0, 0.2):
            setup = SetupBase.find_setup_class("circumbinary-disk")(eccentricity=e, mass_ratio=q)
            for t in times:
                m1, m2 = setup.point_masses(float(t))
                for m in (m1, m2):
                    rows.append([e, q, t, m.position_x, m.position_y, m.velocity_x, m.velocity_y, m.mass,
                                 m.softening_length, m.sink_rate, m.sink_radius, m.sink_model.value])
    return np.array(rows)


def main():
    SetupBase, make_solver = reference_modules()
    run = lambda *a, **k: run_reference(SetupBase, make_solver, *a, **k)
    out = {
        "iso_disk_64": run("circumbinary-disk", 64, 20),
        "iso_disk_128_p4": run("circumbinary-disk", 128, 20, num_patches=4),
        "iso_disk_rk3_48": run("circumbinary-disk", 48, 8, options=dict(rk_order=3), num_patches=3),
        "iso_explosion_64": run("cylindrical-explosion", 64, 20),
        "gam_disk_64": run("circumbinary-disk", 64, 20, model=dict(eos="gamma-law")),
        "gam_cooling_48": run("circumbinary-disk", 48, 10, model=dict(eos="gamma-law", cooling_coefficient=1e-2)),
        "gam_explosion_64": run("cylindrical-explosion", 64, 20, model=dict(eos="gamma-law")),
        # developed flows (spiral shocks, streams into the sinks, the buffer at work), restarted as from a checkpoint
        "iso_disk_96_developed": run("circumbinary-disk", 96, 20, spinup=1500),
        "gam_disk_96_developed": run("circumbinary-disk", 96, 20, model=dict(eos="gamma-law"), spinup=1500, num_patches=2),
    }
    out["iso_reductions_64_p2"] = run_reductions(SetupBase, make_solver, 64, 12, ISO_DIAGNOSTICS,
                                                 model=dict(eccentricity=0.2, mass_ratio=0.5, domain_radius=3.0, sink_radius=0.1),
                                                 num_patches=2)
    out["gam_reductions_48"] = run_reductions(SetupBase, make_solver, 48, 12, GAM_DIAGNOSTICS,
                                              model=dict(eos="gamma-law", constant_softening=False, domain_radius=3.0, mass_ratio=0.4, eccentricity=0.1,
                                                         sink_radius=0.1))
    srhd = lambda *a, **k: run_srhd(SetupBase, make_solver, *a, **k)
    out["srhd_uniform_polar"] = srhd("uniform-polar", 20, 10)
    out["srhd_envelope_shock_p3"] = srhd("envelope-shock", 40, 8, model=dict(polar_extent=0.5), num_patches=3,
                                         options=dict(compute_wavespeed=True))
    out["srhd_envelope_jet_p2"] = srhd("envelope-shock", 40, 8, num_patches=2,
                                       model=dict(polar_extent=0.5, u_shell=0.0, jet_energy=1e-3, jet_theta=0.3))
    out["srhd_envelope_rk3_theta2"] = srhd("envelope-shock", 32, 6, model=dict(polar_extent=0.25, expand=False),
                                           options=dict(rk_order=3, plm_theta=2.0, mach_ceiling=100.0))
    dg = lambda *a, **k: run_dg(SetupBase, make_solver, *a, **k)
    out["dg_disk_32"] = dg("kitp-code-comparison", 32, 10, model=dict(use_dg=True, domain_radius=4.0, sink_radius=0.2))
    out["dg_disk_48_p3_rk3"] = dg("kitp-code-comparison", 48, 6, num_patches=3, options=dict(rk_order=3),
                                  model=dict(use_dg=True, domain_radius=3.0, sink_radius=0.2, eccentricity=0.3,
                                             mass_ratio=0.5, sink_model="torque_free"))
    out["dg_explosion_40_nolimit"] = dg("cylindrical-explosion", 40, 12, model=dict(use_dg=True),
                                        options=dict(limit_slopes=False), num_patches=2)
    out["dg_explosion_40"] = dg("cylindrical-explosion", 40, 12, model=dict(use_dg=True))
    only = set(sys.argv[1:])  # names given on the command line: write those fixtures only (the others are left as they are)
    for name, data in out.items():
        if only and name not in only:
            continue
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
        print(name, data["final"].shape if "final" in data else data["values"][:3],
              "t_end", data["time"][-1] if data["time"].ndim else float(data["time"]), flush=True)
    if not only or "kepler" in only:
        np.savez_compressed(os.path.join(HERE, "kepler.npz"), rows=kepler_fixture(SetupBase))
    # one-patch == four-patch bitwise for the power-of-two mesh (recorded, then relied upon by tests)
    one = run("circumbinary-disk", 128, 20, num_patches=1)
    assert np.array_equal(one["final"], out["iso_disk_128_p4"]["final"])
    print("128^2: 1 patch == 4 patches bitwise")


if __name__ == "__main__":
    main()
