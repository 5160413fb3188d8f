"""A CPIC scene with joints, after scripts/mls-cpic/water_wheel.py: a water wheel built from TWO rigid bodies (four spokes
and four buckets) on a fixed axle — `scripted_position` pins the centres, `rotation_axis` leaves only the spin about z
free — that are tied together by a `rotation` joint (`mpm.add_articulation`, src/articulation.cpp:23-46) and turned by a
stream of water poured onto one side, a new slab of particles every frame.  Needs an MI355X.

    python examples/water_wheel.py [out_dir] [frames] [res]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import taichi_mpm_amd as tc_amd  # noqa: E402


def quad(p0, p1, p2, p3):
    return [[p0, p1, p2], [p0, p2, p3]]


def spokes(r=0.16, w=0.04):
    """four thin rectangular spokes in the x-y plane directions, extruded along z"""
    t = []
    for k in range(4):
        c, s = np.cos(k * np.pi / 2), np.sin(k * np.pi / 2)
        t += quad((0, 0, -w), (r * c, r * s, -w), (r * c, r * s, w), (0, 0, w))
    return np.array(t, np.float32)


def buckets(r=0.16, d=0.05, w=0.04):
    """a small plate at the end of every spoke, perpendicular to it"""
    t = []
    for k in range(4):
        c, s = np.cos(k * np.pi / 2), np.sin(k * np.pi / 2)
        a, b = (r * c - d * s, r * s + d * c), (r * c + d * s, r * s - d * c)
        t += quad((a[0], a[1], -w), (b[0], b[1], -w), (b[0], b[1], w), (a[0], a[1], w))
    return np.array(t, np.float32)


if __name__ == '__main__':
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/water_wheel_frames"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    r = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    mpm = tc_amd.MPM(res=(r, r, r), base_delta_t=5e-5, frame_dt=0.01, num_frames=frames, gravity=(0, -10, 0),
                     frame_directory=out, penalty=1e3, max_particles=1 << 21)
    levelset = mpm.create_levelset()
    levelset.add_plane((0, 1, 0), d=-0.1)
    levelset.set_friction(0.2)
    mpm.set_levelset(levelset, False)
    axle = dict(type='rigid', codimensional=True, density=40, friction=0.2, rotation_axis=(0, 0, 1), angular_damping=3,
                scripted_position=lambda t: (0.5, 0.5, 0.5))  # tc.constant_function13(tc.Vector(0.5, 0.5, 0.5))
    object1 = mpm.add_particles(mesh=spokes(), **axle)
    object2 = mpm.add_particles(mesh=buckets(), **axle)
    mpm.add_articulation(type='rotation', obj0=object1, obj1=object2)

    cells = r // 16  # the stream: a slab of water above the right-hand bucket, falling
    lo = (int(0.62 * r), int(0.80 * r), int(0.5 * r) - cells // 2)

    def frame_update(t, frame_dt):
        mpm.add_particles(type='water', cube_lo=lo, cube_cells=cells, initial_velocity=(0, -0.5, 0))

    mpm.simulate(frame_update=frame_update)
    for rid in (object1, object2):
        st = mpm.c.get_rigid_state(int(rid))
        print("body", rid, "angular velocity", st["angular_velocity"], "rotation", st["rotation"])
    print("frames written to", out, ":", sorted(os.listdir(out))[:3], "...")
