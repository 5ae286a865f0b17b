"""Simulator-backend adapter: the few calls whose spelling differs between simulators.

Only ``full_mass_matrix`` is needed: the reference expands MuJoCo's sparse ``qM`` with
``mjp.cymj._mj_fullM`` (/root/reference/irl_control/robot.py:69).
"""
import numpy as np


def full_mass_matrix(sim, out: np.ndarray) -> np.ndarray:
    """Write the dense nv*nv joint-space inertia matrix of ``sim`` into ``out`` (flat, nv*nv)."""
    if hasattr(sim, "fullM"):                       # FakeSim / any injected backend
        out[:] = np.asarray(sim.fullM(), dtype=np.float64).reshape(-1)
        return out
    try:                                            # official bindings
        import mujoco
        mujoco.mj_fullM(sim.model, out.reshape(sim.model.nv, sim.model.nv), sim.data.qM)
        return out
    except ImportError:
        pass
    import mujoco_py as mjp                         # legacy bindings (what the reference uses)
    mjp.cymj._mj_fullM(sim.model, out, sim.data.qM)
    return out
