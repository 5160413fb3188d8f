"""A CPIC scene in the style of scripts/mls-cpic/sand_paddles.py / sand_stir.py: a scripted paddle wheel stirs a block of
sand resting on a floor.  The rigid body is a thin two-sided shell (`codimensional=True`): the colored distance field
keeps the sand on either side of a blade apart.  One .bgeo frame per frame_dt; the boundary particles of the paddle are in
the frames too (type = 1), as in the reference's output.  Needs an MI355X.

    python examples/sand_paddle.py [out_dir] [frames]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import taichi_mpm_amd as tc_amd  # noqa: E402


def paddle(r=0.16, h=0.10):
    """two crossed rectangular blades around the z axis"""
    a = np.array([[[-r, -h, 0], [r, -h, 0], [r, h, 0]], [[-r, -h, 0], [r, h, 0], [-r, h, 0]]], np.float32)
    return np.concatenate([a, a[:, :, [2, 1, 0]]])


if __name__ == '__main__':
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/sand_paddle_frames"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    r = 128
    mpm = tc_amd.MPM(res=(r, r, r), base_delta_t=1e-4, frame_dt=0.01, num_frames=frames, gravity=(0, -10, 0),
                     frame_directory=out, penalty=1e4, max_particles=1 << 21)
    levelset = mpm.create_levelset()
    levelset.add_plane((0, 1, 0), d=-0.2)
    levelset.set_friction(-1)
    mpm.set_levelset(levelset, False)
    rid = mpm.add_particles(type='rigid', mesh=paddle(), codimensional=True, friction=-2, density=500,
                            scripted_position=lambda t: (0.5, 0.34, 0.5),          # tc.constant_function13(...)
                            scripted_rotation=lambda t: (0.0, 360.0 * t, 0.0))     # one turn per second about y
    mpm.add_particles(type='sand', cube=(40, 88), friction_angle=30, initial_velocity=(0, 0, 0))
    mpm.simulate()
    print("rigid body", rid, "state:", mpm.c.get_rigid_state(int(rid))["rotation"])
    print("frames written to", out, ":", sorted(os.listdir(out))[:3], "...")
