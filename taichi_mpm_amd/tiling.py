"""Jobs that bench.py times: one GPU, or the grid tiled across the GPUs of one node (one process per GPU).

Round-1 state: the single-GPU job is complete; the multi-GPU job is built in `taichi_mpm_amd/tiled.py`
(slab decomposition + halo sum + particle migration over torch.distributed/RCCL)."""


class SingleJob:
    scaling = "strong"
    parallelism = "1 GPU"

    def __init__(self, sim):
        self.sim = sim
        sim._ensure_ctx()

    def num_particles(self):
        return self.sim.get_num_particles()

    def run(self, n):
        self.sim.run_substeps(n)

    def synchronize(self):
        self.sim.synchronize()

    def set_profiling(self, level):
        self.sim.set_profiling(level)
        self.sim.profile(reset=True)

    def profile(self):
        return self.sim.profile()


def make_job(tm, cfg, rank, world, local_rank, build_sim):
    if world == 1:
        return SingleJob(build_sim(tm, cfg, local_rank))
    from . import tiled
    return tiled.make_tiled_job(tm, cfg, rank, world, local_rank)
