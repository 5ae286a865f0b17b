"""Simulator-backend adapter: the few calls whose spelling differs between simulators.

Only ``full_mass_matrix`` is needed: the reference expands MuJoCo's sparse ``qM`` with
``mjp.cymj._mj_fullM`` (/root/reference/irl_control/robot.py:69).  The backend is chosen by what
``sim`` IS, never by which packages happen to be importable: an injected backend (FakeSim, the
MujocoSim adapter of mujoco_backend.py) brings its own ``fullM``; a ``mujoco_py.MjSim`` (what the
reference's ``MujocoApp(scene_file=...)`` creates) goes to mujoco_py even when the official
bindings are installed next to it.
"""
import numpy as np


def backend_of(sim) -> str:
    """'injected' (has fullM), 'mujoco' (official bindings' MjModel/MjData pair) or 'mujoco_py'."""
    if hasattr(sim, "fullM"):
        return "injected"
    mod = type(getattr(sim, "model", None)).__module__ or ""
    if mod.split(".")[0] == "mujoco":
        return "mujoco"
    return "mujoco_py"


def full_mass_matrix(sim, out: np.ndarray) -> np.ndarray:
    """Write the dense nv*nv joint-space inertia matrix of ``sim`` into ``out`` (flat, nv*nv)."""
    kind = backend_of(sim)
    if kind == "injected":
        out[:] = np.asarray(sim.fullM(), dtype=np.float64).reshape(-1)
    elif kind == "mujoco":
        import mujoco
        mujoco.mj_fullM(sim.model, out.reshape(sim.model.nv, sim.model.nv), sim.data.qM)
    else:
        import mujoco_py as mjp                     # legacy bindings (what the reference uses)
        mjp.cymj._mj_fullM(sim.model, out, sim.data.qM)
    return out
