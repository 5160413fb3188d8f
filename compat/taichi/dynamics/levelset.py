"""`from taichi.dynamics.levelset import LevelSet` (scripts/async/async_mpm.py:2): `LevelSet(Vectori(res), Vector(0.0))`,
`.levelset` = the core object the driver hands to DynamicLevelSet.initialize (:119-127)."""
from taichi_mpm_amd.mpm import LevelSet as _LevelSet


class LevelSet(_LevelSet):
    def __init__(self, res=None, offset=None, friction=-1.0, delta_x=None):
        if delta_x is None and res is not None and len(res):
            delta_x = 1.0 / float(res[0])
        super().__init__(friction=friction, delta_x=delta_x)

    @property
    def levelset(self):
        return self
