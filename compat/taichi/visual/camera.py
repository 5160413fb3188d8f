"""inert: imported by the reference's driver (scripts/async/async_mpm.py:5), used only for rendering"""


class Camera:
    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k
