"""inert: imported by the reference's driver (scripts/async/async_mpm.py:6), used only for rendering"""


class ParticleRenderer:
    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k

    def set_camera(self, camera):
        return None

    def render(self, *a, **k):
        return None
