"""``Device``: one controlled kinematic chain (base yaw, right arm, left arm) of a robot.

API and behaviour follow /root/reference/irl_control/device.py:7-213 — same constructor
signature, attribute names, ``DeviceState`` keys and getter semantics — because the examples and
``OSC`` address devices through exactly these names.  The state readers are a table of
(enum -> reader) rather than lambdas, and the per-tick path used by the batched controller is
``pack_pose7`` / ``pack_wrench6`` / ``jac6`` which return the C-ABI record layouts directly.
"""
import copy
from enum import Enum
from threading import Lock
from typing import Any, Dict

import numpy as np


class DeviceState(Enum):
    Q = 'Q'
    Q_ACTUATED = 'Q_ACTUATED'
    DQ = 'DQ'
    DQ_ACTUATED = 'DQ_ACTUATED'
    DDQ = 'DDQ'
    EE_XYZ = 'EE_XYZ'
    EE_XYZ_VEL = 'EE_XYZ_VEL'
    EE_QUAT = 'EE_QUAT'
    FORCE = 'FORCE'
    TORQUE = 'TORQUE'
    J = 'JACOBIAN'


# F/T sensor slices of sensordata per device name and the site whose frame they are measured in
# (reference device.py:139-170 hard-codes these by name; dual_ur5.xml:289-297 defines the order).
_FT_TABLE = {
    "ur5right": ("ft_frame_ur5right", slice(0, 3), slice(3, 6)),
    "ur5left": ("ft_frame_ur5left", slice(6, 9), slice(9, 12)),
}


def _chain_joints(model, ee_body: int, stop_body: int):
    """Walk EE -> root collecting joint ids until the parent is the world or ``stop_body``
    (reference device.py:49-64).  Returned base-first."""
    ids, names = [], []
    b = ee_body
    while model.body_parentid[b] != 0 and model.body_parentid[b] != stop_body:
        adr, num = int(model.body_jntadr[b]), int(model.body_jntnum[b])
        for j in range(adr + num - 1, adr - 1, -1):
            ids.append(j)
            names.append(model.joint_id2name(j))
        b = model.body_parentid[b]
    return ids[::-1], names[::-1]


class Device():
    def __init__(self, device_yml: Dict, model, sim, use_sim: bool):
        self.sim = sim
        self.__use_sim = use_sim
        y = device_yml
        self.name = y['name']
        self.max_vel = y.get('max_vel')
        self.EE = y['EE']
        self.ctrlr_dof_xyz = y['ctrlr_dof_xyz']
        self.ctrlr_dof_abg = y['ctrlr_dof_abg']
        self.ctrlr_dof = np.hstack([self.ctrlr_dof_xyz, self.ctrlr_dof_abg])
        self.start_angles = np.array(y['start_angles'])
        self.num_gripper_joints = y['num_gripper_joints']

        try:
            stop_body = model.body_name2id(y['start_body'])
        except Exception:
            stop_body = 0
        jids, jnames = _chain_joints(model, model.body_name2id(self.EE), stop_body)
        self.joint_names = jnames
        self.joint_ids = np.array(jids)
        g0 = self.joint_ids[-1] + 1
        self.gripper_ids = np.arange(g0, g0 + self.num_gripper_joints)
        self.joint_ids_all = np.hstack([self.joint_ids, self.gripper_ids])

        trn = model.actuator_trnid[:, 0]
        self.ctrl_idxs = np.intersect1d(trn, self.joint_ids_all, return_indices=True)[1]
        self.actuator_trnids = trn[self.ctrl_idxs]

        if self.name in ("ur5right", "ur5left", "base"):
            # raises ValueError on a length mismatch exactly like the reference (device.py:76-79)
            self.sim.data.qpos[self.joint_ids] = np.copy(self.start_angles)
        self.sim.forward()

        if np.sum(self.ctrlr_dof) > len(self.joint_ids):
            print("Fewer DOF than specified")

        d = self.sim.data
        self.__readers = {
            DeviceState.Q: lambda: d.qpos[self.joint_ids_all],
            DeviceState.Q_ACTUATED: lambda: d.qpos[self.joint_ids],
            DeviceState.DQ: lambda: d.qvel[self.joint_ids_all],
            DeviceState.DQ_ACTUATED: lambda: d.qvel[self.joint_ids],
            DeviceState.DDQ: lambda: d.qacc[self.joint_ids_all],
            DeviceState.EE_XYZ: lambda: d.get_body_xpos(self.EE),
            DeviceState.EE_XYZ_VEL: lambda: d.get_body_xvelp(self.EE),
            DeviceState.EE_QUAT: lambda: d.get_body_xquat(self.EE),
            DeviceState.FORCE: lambda: self._wrench_part(1),
            DeviceState.TORQUE: lambda: self._wrench_part(2),
            DeviceState.J: lambda: self.jac6()[self.ctrlr_dof],
        }
        self.__state: Dict[DeviceState, Any] = dict()
        self.__state_locks: Dict[DeviceState, Lock] = {k: Lock() for k in DeviceState}
        self.concise_state_vars = [
            DeviceState.Q_ACTUATED, DeviceState.DQ_ACTUATED, DeviceState.EE_XYZ,
            DeviceState.EE_XYZ_VEL, DeviceState.EE_QUAT, DeviceState.FORCE, DeviceState.TORQUE,
        ]

    # ---- raw readers --------------------------------------------------------------------------
    def jac6(self) -> np.ndarray:
        """Unmasked 6 x nv EE Jacobian: jacp rows then jacr rows (reference device.py:123-130)."""
        d = self.sim.data
        return np.vstack([np.asarray(d.get_body_jacp(self.EE)).reshape(3, -1),
                          np.asarray(d.get_body_jacr(self.EE)).reshape(3, -1)])

    def _wrench_part(self, which: int) -> np.ndarray:
        """F (which=1) or tau (which=2) rotated into the world by the F/T site frame; zeros for
        devices without a sensor (reference device.py:135-170)."""
        ent = _FT_TABLE.get(self.name)
        if ent is None:
            return np.zeros(3)
        R = self.sim.data.get_site_xmat(ent[0])
        return np.matmul(np.asarray(R).reshape(3, 3), self.sim.data.sensordata[ent[which]])

    # ---- reference getters ----------------------------------------------------------------------
    def get_state(self, state_var: DeviceState):
        if self.__use_sim:
            return copy.copy(self.__readers[state_var]())
        with self.__state_locks[state_var]:
            return copy.copy(self.__state[state_var])

    def get_all_states(self):
        return {key: self.get_state(key) for key in self.concise_state_vars}

    def update_state(self):
        """Polling-thread body (Robot.start); only legal when not reading the sim directly."""
        assert self.__use_sim is False
        for var in DeviceState:
            with self.__state_locks[var]:
                self.__state[var] = copy.copy(self.__readers[var]())

    def get_all_joint_ids(self):
        return self.joint_ids_all

    def get_actuator_joint_ids(self):
        return self.joint_ids

    def get_gripper_joint_ids(self):
        return self.gripper_ids

    # ---- packed records for the C ABI -----------------------------------------------------------
    def pack_pose7(self) -> np.ndarray:
        return np.concatenate([np.asarray(self.get_state(DeviceState.EE_XYZ), dtype=np.float64),
                               np.asarray(self.get_state(DeviceState.EE_QUAT), dtype=np.float64)])

    def pack_wrench6(self) -> np.ndarray:
        return np.concatenate([self.get_state(DeviceState.FORCE),
                               self.get_state(DeviceState.TORQUE)]).astype(np.float64)
