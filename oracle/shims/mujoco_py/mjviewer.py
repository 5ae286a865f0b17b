"""Stub of mujoco_py.mjviewer for importing the reference's example scripts (examples/insertion_task.py:1)."""


class MjViewer:
    def __init__(self, sim=None):
        self.sim = sim
        self.cam = type("Cam", (), {})()
        self.on_render = None

    def render(self):
        if self.on_render is not None:
            self.on_render()
