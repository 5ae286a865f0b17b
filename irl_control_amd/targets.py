"""``Target`` and ``ControllerConfig``: the value types handed to ``OSC.generate``.

API parity with /root/reference/irl_control/utils.py:5-80 (same method names, argument meaning and
assertion behaviour; quaternions are stored w,x,y,z).  Internally a target is kept as two packed
7-vectors ``[x y z qw qx qy qz]`` (pose and "velocity pose") because that is the record layout the
C ABI consumes (include/irlosc.h: ``tgt_pose[B,ndev,7]``).
"""
from typing import Any, Sequence

import numpy as np

from .transforms import euler2quat, quat2euler


def _vec(v, n):
    assert len(v) == n
    return np.asarray(v)


class Target:
    __slots__ = ("_xyz", "_quat", "_xyz_vel", "_quat_vel")

    def __init__(self, xyz_abg: Sequence = np.zeros(6), xyz_abg_vel: Sequence = np.zeros(6)):
        assert len(xyz_abg) == 6 and len(xyz_abg_vel) == 6
        pose, vel = np.array(xyz_abg), np.array(xyz_abg_vel)
        self._xyz, self._xyz_vel = pose[:3], vel[:3]
        self._quat = np.array(euler2quat(*pose[3:]))
        self._quat_vel = np.array(euler2quat(*vel[3:]))

    # ---- getters (utils.py:17-33) -------------------------------------------------------------
    def get_xyz(self): return self._xyz
    def get_xyz_vel(self): return self._xyz_vel
    def get_quat(self): return self._quat
    def get_quat_vel(self): return np.asarray(self._quat_vel)
    def get_abg(self): return np.asarray(quat2euler(self._quat))
    def get_abg_vel(self): return np.asarray(quat2euler(self._quat_vel))

    # ---- setters (utils.py:35-67) -------------------------------------------------------------
    def set_xyz(self, xyz): self._xyz = _vec(xyz, 3)
    def set_xyz_vel(self, xyz_vel): self._xyz_vel = _vec(xyz_vel, 3)
    def set_quat(self, quat): self._quat = _vec(quat, 4)
    def set_quat_vel(self, quat_vel): self._quat_vel = _vec(quat_vel, 4)
    def set_abg(self, abg): self._quat = np.asarray(euler2quat(*_vec(abg, 3)))
    def set_abg_vel(self, abg_vel): self._quat_vel = np.asarray(euler2quat(*_vec(abg_vel, 3)))

    def set_all_quat(self, xyz, quat):
        assert len(xyz) == 3 and len(quat) == 4
        self._xyz, self._quat = np.asarray(xyz), np.asarray(quat)

    def set_all_abg(self, xyz, abg):
        assert len(xyz) == 3 and len(abg) == 3
        self._xyz, self._quat = np.asarray(xyz), np.asarray(euler2quat(*abg))

    # ---- packed records for the C ABI ---------------------------------------------------------
    def pose7(self) -> np.ndarray:
        """[x y z qw qx qy qz] (float64)."""
        return np.concatenate([np.asarray(self._xyz, dtype=np.float64),
                               np.asarray(self._quat, dtype=np.float64)])

    def vel6(self) -> np.ndarray:
        """hstack(xyz_vel, abg_vel) exactly as /root/reference/irl_control/osc.py:172 forms it."""
        return np.hstack([self.get_xyz_vel(), self.get_abg_vel()]).astype(np.float64)


class ControllerConfig:
    """dict wrapper for one gain set; ``OSC.__init__`` adds 'task_space_gains' and 'lamb' to it."""

    def __init__(self, ctrlr_dict):
        self.ctrlr_dict = ctrlr_dict

    def __getitem__(self, key: str) -> Any:
        return self.ctrlr_dict[key]

    def __setitem__(self, key: str, value: Any) -> None:
        self.ctrlr_dict[key] = value

    def get_params(self, keys):
        return [self.ctrlr_dict[k] for k in keys]
