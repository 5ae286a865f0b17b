"""``Robot``: groups Devices and assembles the joint-space inputs of the OSC path
(M, dq, per-device Jacobians).  API parity with /root/reference/irl_control/robot.py:11-144.
"""
import copy
import time
from enum import Enum
from threading import Lock
from typing import Any, Dict, List

import numpy as np

from .backend import full_mass_matrix
from .device import Device, DeviceState


class RobotState(Enum):
    M = 'INERTIA'
    DQ = 'DQ'
    J = 'JACOBIAN'


class Robot():
    def __init__(self, sub_devices: List[Device], robot_name, sim, use_sim, collect_hz=1000):
        self.sim = sim
        self.__use_sim = use_sim
        self.sub_devices = sub_devices
        self.sub_devices_dict: Dict[str, Device] = {dev.name: dev for dev in sub_devices}
        self.name = robot_name
        self.num_scene_joints = self.sim.model.nv
        self.M_vec = np.zeros(self.num_scene_joints ** 2)
        ids = np.array([], dtype=np.int32)
        for dev in self.sub_devices:
            ids = np.hstack([ids, dev.joint_ids_all])
        self.joint_ids_all = np.sort(np.unique(ids))
        self.num_joints_total = len(self.joint_ids_all)
        self.running = False
        self.data_collect_hz = collect_hz
        self.__assemble = {RobotState.M: self._assemble_M, RobotState.DQ: self._assemble_dq,
                           RobotState.J: self._assemble_J}
        self.__state: Dict[RobotState, Any] = dict()
        self.__state_locks: Dict[RobotState, Lock] = {k: Lock() for k in RobotState}

    # ---- assembly (robot.py:44-72) -------------------------------------------------------------
    def _assemble_J(self):
        """({name: J_d[r_d, n]}, {name: row index range}) over ALL sub-devices in their order."""
        Js, J_idxs, row = {}, {}, 0
        for name, device in self.sub_devices_dict.items():
            Jd = device.get_state(DeviceState.J)
            J_idxs[name] = np.arange(row, row + Jd.shape[0])
            row += Jd.shape[0]
            Js[name] = Jd[:, self.joint_ids_all]
        return Js, J_idxs

    def _assemble_dq(self):
        dq = np.zeros(self.joint_ids_all.shape)
        for dev in self.sub_devices:
            dq[dev.get_all_joint_ids()] = dev.get_state(DeviceState.DQ)   # raw ids as positions
        return dq

    def _assemble_M(self):
        full_mass_matrix(self.sim, self.M_vec)
        nv = self.num_scene_joints
        return self.M_vec.reshape(nv, nv)[np.ix_(self.joint_ids_all, self.joint_ids_all)]

    # ---- reference getters -----------------------------------------------------------------------
    def get_state(self, state_var: RobotState):
        if self.__use_sim:
            return copy.copy(self.__assemble[state_var]())
        with self.__state_locks[state_var]:
            return copy.copy(self.__state[state_var])

    def get_all_states(self):
        state = self.get_device_states()
        for key in RobotState:
            state[key] = self.get_state(key)
        return state

    def get_device_states(self):
        return {name: dev.get_all_states() for name, dev in self.sub_devices_dict.items()}

    def get_device(self, device_name: str) -> Device:
        return self.sub_devices_dict[device_name]

    def is_running(self):
        return self.running

    def is_using_sim(self):
        return self.__use_sim

    # ---- polling thread (robot.py:98-120); unused by the batched path, kept for API parity ------
    def __refresh(self):
        assert self.__use_sim is False
        for var in RobotState:
            with self.__state_locks[var]:
                self.__state[var] = copy.copy(self.__assemble[var]())

    def start(self):
        assert self.running is False and self.__use_sim is False
        self.running = True
        period = 1.0 / float(self.data_collect_hz)
        last = time.time()
        while self.running:
            for dev in self.sub_devices:
                dev.update_state()
            self.__refresh()
            now = time.time()
            time.sleep(max(period - (now - last), 0))
            last = now

    def stop(self):
        assert self.running is True and self.__use_sim is False
        self.running = False
