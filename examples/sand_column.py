"""A small scene in the style of the reference's scene scripts (scripts/mls-cpic/*.py, scripts/async/sand.py): a sand
column collapsing on a floor inside a box, one .bgeo frame per frame_dt for Houdini — the files are byte-compatible
with the reference's Partio output.  Needs an MI355X.

    python examples/sand_column.py [out_dir] [frames]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import taichi_mpm_amd as tc_amd  # noqa: E402

if __name__ == '__main__':
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/sand_column_frames"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    r = 128
    mpm = tc_amd.MPM(res=(r, r, r), base_delta_t=1e-4, frame_dt=0.01, num_frames=frames, gravity=(0, -10, 0),
                     frame_directory=out, verbose_bgeo=False)
    levelset = mpm.create_levelset()
    levelset.add_plane((0, 1, 0), d=-0.1)                      # floor y = 0.1 (scripts/async/sand.py:34-37)
    levelset.add_cuboid((0.1, 0.0, 0.1), (0.9, 1.0, 0.9), True)  # the box around it
    levelset.set_friction(0.4)
    mpm.set_levelset(levelset, False)
    mpm.add_particles(type='sand', cube=(54, 74), friction_angle=30, initial_velocity=(0, -1, 0))
    mpm.simulate()
    print("frames written to", out, ":", sorted(os.listdir(out))[:3], "...")
