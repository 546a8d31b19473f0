"""
Initial data made on the device (sfhip_circumbinary_disk_primitive) against the
reference's per-point formulas (CircumbinaryDisk.primitive, circumbinary_disk.py:82-93, restated in
sailfish_amd/setups.py and pinned bitwise by tests/test_golden_cpu.py): bit-identical, for every patch of a
decomposition, on meshes whose spacing is and is not a power of two.
"""
import numpy as np
import pytest

from sailfish_amd.setups import make_setup
from sailfish_amd.solvers import make_solver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("resolution,num_patches,model", [
    (128, 1, {}), (100, 3, {}), (257, 4, dict(softening_length=0.013, initial_sigma=0.7, domain_radius=3.3)),
])
def test_device_initial_data_equal_the_host_formulas(resolution, num_patches, model, monkeypatch):
    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(resolution)
    expect = setup.primitive_array(0.0, mesh, 0, resolution, 3)
    point = [0.0] * 3  # and the vectorised host form is the per-point loop (spot check, all fields)
    for i, j in ((0, 0), (resolution // 2, resolution // 2), (resolution - 1, 3)):
        setup.primitive(0.0, mesh.cell_coordinates(i, j), point)
        assert list(expect[i, j]) == point
    monkeypatch.delenv("SFHIP_HOST_IC", raising=False)
    made_on_device = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0,
                                 num_patches=num_patches, mode="hip")
    assert np.array_equal(made_on_device.solution, expect)
    for p in made_on_device.patches:  # guards untouched (zero), the other buffers start as copies
        assert float(p.primitive1[:2].abs().max()) == 0.0 and float(p.primitive1[:, -2:].abs().max()) == 0.0
        assert bool((p.primitive2 == p.primitive1).all())
    monkeypatch.setenv("SFHIP_HOST_IC", "1")
    copied = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, num_patches=num_patches,
                         mode="hip")
    assert np.array_equal(copied.solution, expect)
    # same state, same first step, bit for bit
    assert made_on_device.step(0.3) == copied.step(0.3)
    assert np.array_equal(made_on_device.solution, copied.solution)


def test_gamma_law_disk_stays_on_the_host():
    setup = make_setup("circumbinary-disk", eos="gamma-law")
    assert not setup.has_device_primitive
    mesh = setup.mesh(64)
    s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
    assert np.array_equal(s.solution, setup.primitive_array(0.0, mesh, 0, 64, 4))
