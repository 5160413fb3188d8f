"""Oracle level set (planes + spheres + cuboids) and particle_collision_resolution (src/mpm.cpp:313-368, 414-426)."""
import numpy as np

from tests.common import lattice_cube, make_state

RES, DX, DT = 32, 1.0 / 32, 1e-4


def test_sphere_and_cuboid_boundary_conditions_act_on_the_right_nodes(orc):
    """a node inside a sticky solid (within 3 cells of its surface) is stopped; outside or deeper it is untouched"""
    shape = (RES + 1,) * 3 + (4,)
    for shapes, probe_in, probe_deep, probe_out in (
        ([(1, 0, 0.5, 0.5, 0.5, 0.25)], (16, 16, 9), (16, 16, 16), (16, 16, 4)),          # solid ball r = 8 cells
        ([(2, 0, 0.25, 0.25, 0.25, 0.75, 0.75, 0.75)], (9, 16, 16), (16, 16, 16), (5, 16, 16)),  # solid box 8..24
        ([(2, 1, 0.25, 0.25, 0.25, 0.75, 0.75, 0.75)], (7, 16, 16), (2, 16, 16), (12, 16, 16)),  # container
    ):
        cfg = orc.make_config(RES, DX, DT, friction=-1.0, shapes=shapes, particle_gravity=True)
        g = np.zeros(shape, np.float32)
        g[..., 3] = 1.0
        g[..., :3] = (1.0, 2.0, 3.0)  # momentum (mass 1 => velocity)
        out = orc.grid_update(cfg, g.copy())
        assert np.allclose(out[probe_in][:3], 0.0), shapes
        assert np.allclose(out[probe_deep][:3], (1, 2, 3)), shapes   # deeper than 3 cells: skipped (:324-325)
        assert np.allclose(out[probe_out][:3], (1, 2, 3)), shapes


def test_particle_collision_projects_onto_the_surface(orc):
    x = np.array([[0.5, 0.28, 0.5], [0.5, 0.5, 0.5], [0.62, 0.5, 0.5]], np.float32)
    s = make_state(x, "jelly", DX)
    s.v[:] = [(0, -1, 0), (0, -1, 0), (-2, 0.5, 0)]
    cfg = orc.make_config(RES, DX, DT, planes=[(0, 1, 0, -0.3)], shapes=[(1, 0, 0.7, 0.5, 0.5, 0.1)], particle_collision=True)
    orc.particle_collision(cfg, s)
    assert np.allclose(s.x[0], (0.5, 0.3, 0.5), atol=1e-6) and np.allclose(s.v[0], 0, atol=1e-6)   # below the floor
    assert np.allclose(s.x[1], (0.5, 0.5, 0.5)) and np.allclose(s.v[1], (0, -1, 0))                # free space
    assert np.allclose(s.x[2], (0.6, 0.5, 0.5), atol=1e-6) and np.allclose(s.v[2], (0, 0.5, 0), atol=1e-6)  # inside the ball


def test_substep_with_container_keeps_particles_inside(orc):
    x = lattice_cube(RES, 10, 16, DX, jitter=0.2, seed=3)
    s = make_state(x, "water", DX, vel_scale=0.0)
    s.v[:] = (3.0, -2.0, 0.0)
    box = (2, 1, 0.3, 0.3, 0.3, 0.6, 0.6, 0.6)  # container around the block
    cfg = orc.make_config(RES, DX, 2e-4, friction=-2.0, shapes=[box], particle_collision=True)
    for _ in range(40):
        orc.substep(cfg, s)
    assert s.n == len(x)
    assert (s.x >= 0.3 - 1e-5).all() and (s.x <= 0.6 + 1e-5).all()
