"""`tc.dynamics.MPM` — the scene-script driver (scripts/benchmark/benchmark_3d.py:9-18; in-tree twin of its shape:
scripts/async/async_mpm.py:17-300)."""
import os

from taichi_mpm_amd.mpm import MPM as _MPM

from . import levelset  # noqa: F401


class MPM(_MPM):
    """taichi_mpm_amd.MPM with the legacy defaults a verbatim script relies on: frames go under tc.get_output_path (the legacy
    driver always had an output directory), and TAICHI_MPM_NUM_FRAMES bounds the frame loop from outside the script."""

    def __init__(self, snapshot_interval=20, **kwargs):
        if "output_directory" not in kwargs and "frame_directory" not in kwargs and os.environ.get("TAICHI_MPM_OUTPUT"):
            import taichi
            task = kwargs.get("task_id", "mpm")
            kwargs["output_directory"] = taichi.get_output_path(os.path.join("mpm", str(task)), True)
        super().__init__(snapshot_interval=snapshot_interval, **kwargs)
        cap = os.environ.get("TAICHI_MPM_NUM_FRAMES")
        if cap:
            self.num_frames = min(int(self.num_frames), int(cap))
