"""inert: imported by the reference's driver (scripts/async/async_mpm.py:8), used only for rendering"""


def show_image(*a, **k):
    return None
