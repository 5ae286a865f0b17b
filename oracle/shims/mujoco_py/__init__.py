"""Stub of the mujoco_py names the reference imports (osc.py:3, robot.py:3,69, mujoco_app.py:2)."""
import numpy as np


class _Cymj:
    @staticmethod
    def _mj_fullM(model, M_vec, qM):
        M_vec[:] = np.asarray(qM, dtype=np.float64).reshape(-1)


cymj = _Cymj()


def load_model_from_path(path):
    raise RuntimeError("mujoco_py stub: no MuJoCo here; inject a FakeSim instead")


class MjSim:
    def __init__(self, *a, **k):
        raise RuntimeError("mujoco_py stub: no MuJoCo here; inject a FakeSim instead")


class GlfwContext:                       # examples/space_mouse_example.py:1 imports it (never constructed on the headless path)
    def __init__(self, *a, **k):
        raise RuntimeError("mujoco_py stub: no OpenGL context here")
