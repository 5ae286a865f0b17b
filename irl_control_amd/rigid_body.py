"""``RigidBodyModel``: the body table of a robot (tree of hinge joints) for the GPU front end.

The reference never holds such a table: it reads M, the EE Jacobians, qfrc_bias and the EE poses from MuJoCo every
tick (/root/reference/irl_control/robot.py:68-72, device.py:97-99,115-133, osc.py:190-191).  The front end of
libirlosc (irlosc_set_model / irlosc_frontend, csrc/osc_frontend.hpp) computes them from (qpos, qvel) for a whole
batch; this class loads the table (``models/dual_ur5.json``, extracted from the reference's MJCF scene by
tools/parse_mjcf.py) and packs it into ``struct irlosc_model``.
"""
import ctypes as C
import json
import os
from typing import Dict, List

import numpy as np

from . import _lib

MODEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")
# EE body of each device in the Dual-UR5 configs (robot_configs/*.yaml "EE" keys)
DUAL_UR5_EE = {"base": "ur_stand_dummy", "ur5right": "ur_EE_ur5right", "ur5left": "ur_EE_ur5left"}


class RigidBodyModel:
    def __init__(self, table: Dict):
        self.table = table
        self.bodies = table["bodies"]
        self.nb = len(self.bodies)
        self.joint_body = [i for i, b in enumerate(self.bodies) if b["joint"]]
        self.nj = len(self.joint_body)
        if self.nb > _lib.MAX_BODIES or self.nj > _lib.MAX_N:
            raise ValueError(f"{self.nb} bodies / {self.nj} hinges exceed the C ABI limits")

    @classmethod
    def load(cls, name: str = "dual_ur5") -> "RigidBodyModel":
        path = name if os.path.isabs(name) else os.path.join(MODEL_DIR, name + ".json")
        with open(path) as f:
            return cls(json.load(f))

    @property
    def joint_names(self) -> List[str]:
        return [self.bodies[b]["joint"]["name"] for b in self.joint_body]

    @property
    def joint_ranges(self) -> np.ndarray:
        return np.array([self.bodies[b]["joint"]["range"] for b in self.joint_body])

    def body_id(self, name: str) -> int:
        for i, b in enumerate(self.bodies):
            if b["name"] == name:
                return i
        raise KeyError(name)

    def to_struct(self, ee_bodies: List[str]) -> "_lib.Model":
        """``struct irlosc_model`` with ee_body[d] = the body of target device d (targets order)."""
        m = _lib.Model()
        m.nb, m.nj = self.nb, self.nj
        j = 0
        for i, b in enumerate(self.bodies):
            m.parent[i] = b["parent"]
            m.joint_of_body[i] = -1
            for a in range(3):
                m.pos[i][a] = b["pos"][a]; m.ipos[i][a] = b["ipos"][a]; m.inertia[i][a] = b["inertia"][a]
            for a in range(4):
                m.quat[i][a] = b["quat"][a]; m.iquat[i][a] = b["iquat"][a]
            m.mass[i] = b["mass"]
            if b["joint"]:
                m.joint_of_body[i] = j
                for a in range(3):
                    m.jaxis[j][a] = b["joint"]["axis"][a]; m.jpos[j][a] = b["joint"]["pos"][a]
                m.armature[j] = b["joint"].get("armature", 0.0)
                j += 1
        for a in range(3):
            m.gravity[a] = self.table["gravity"][a]
        if len(ee_bodies) > _lib.MAX_DEV:
            raise ValueError("too many target devices")
        for d, name in enumerate(ee_bodies):
            m.ee_body[d] = self.body_id(name)
        return m

    def random_state(self, rng: np.random.Generator, B: int, vel_scale: float = 0.5):
        """(qpos[B,nj], qvel[B,nj]): joints with a range narrower than 3 rad (the gripper) inside it, the rest in [-pi, pi]."""
        r = self.joint_ranges
        narrow = (r[:, 1] - r[:, 0]) < 3.0
        lo = np.where(narrow, r[:, 0], -np.pi)
        hi = np.where(narrow, r[:, 1], np.pi)
        return rng.uniform(lo, hi, size=(B, self.nj)), rng.normal(0.0, vel_scale, size=(B, self.nj))
