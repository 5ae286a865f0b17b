"""``OSC``: operational-space / admittance controller with the reference's call surface
(/root/reference/irl_control/osc.py:12-210) whose numerical body runs in the HIP kernels of
libirlosc (irl_control_amd/csrc/).

``generate(targets)`` keeps the reference's contract — ordered dict of device name -> Target in,
``(force_idxs, forces)`` out, consumed as ``sim.data.ctrl[idx] = force`` — and its quirks
(targets order defines the Jacobian row order and the output order, osc.py:136-138,203-208; the
gain dicts passed to the constructor are mutated in place, osc.py:35-39).  Per call it
  1. reads the state exactly like the reference (robot.get_all_states(), osc.py:132),
  2. packs it into the C-ABI records (B = 1), and
  3. lets the GPU do everything from osc.py:144 to osc.py:200.
There is no NumPy fallback for step 3.
"""
from typing import Dict, List, Tuple

import numpy as np

from . import _lib
from .batched import BatchedOSC
from .device import Device, DeviceState
from .layout import OSCLayout
from .robot import Robot, RobotState
from .targets import ControllerConfig, Target
from .transforms import normalized_vector, qconjugate, qmult, quat2euler


class OSC():
    def __init__(self, robot: Robot, sim, input_device_configs: Tuple[str, Dict],
                 nullspace_config: Dict = None, use_g=True, admittance=False,
                 dtype=np.float64, hip_device: int = 0):
        self.sim = sim
        self.robot = robot
        self.device_configs: Dict[str, ControllerConfig] = dict()
        for name, gains in input_device_configs:
            self.device_configs[name] = ControllerConfig(gains)
        self.nullspace_config = nullspace_config
        self.use_g = use_g
        self.admittance = admittance
        for cc in self.device_configs.values():
            kv, kp, ko = cc.get_params(['kv', 'kp', 'ko'])
            tsg = np.array([kp] * 3 + [ko] * 3)
            cc['task_space_gains'] = tsg
            cc['lamb'] = tsg / kv
        self._dtype = dtype
        self._hip_device = hip_device
        self._ctx: Dict[tuple, BatchedOSC] = {}
        self._layouts: Dict[tuple, OSCLayout] = {}
        self.last_flags = 0

    # ------------------------------------------------------------------------------------------
    def calc_error(self, target: Target, device: Device):
        """EE-minus-target error [xyz, sxyz-Euler] (osc.py:101-118).  Public diagnostic used by
        callers between ticks (examples/insertion_task.py:173-179); inside ``generate`` the same
        quantity is computed on the GPU (csrc/osc_common.hpp task_error6)."""
        e = np.zeros(6)
        if np.sum(device.ctrlr_dof_xyz) > 0:
            e[:3] = device.get_state(DeviceState.EE_XYZ) - target.get_xyz()
        if np.sum(device.ctrlr_dof_abg) > 0:
            q_d = normalized_vector(target.get_quat())
            q_r = np.array(qmult(q_d, qconjugate(device.get_state(DeviceState.EE_QUAT))))
            e[3:] = quat2euler(qconjugate(q_r))
        return e

    # ------------------------------------------------------------------------------------------
    def _layout_for(self, names: List[str], J_idxs) -> OSCLayout:
        devs = [self.robot.get_device(nm) for nm in names]
        # the layout only changes when the caller re-masks a device (ps_move example) or passes other targets:
        # key on exactly what it is built from and reuse the object (and with it the GPU context)
        key = (tuple(names), tuple(tuple(bool(x) for x in dv.ctrlr_dof) for dv in devs),
               tuple((bool(np.sum(dv.ctrlr_dof_xyz) > 0), bool(np.sum(dv.ctrlr_dof_abg) > 0)) for dv in devs),
               tuple(dv.max_vel is not None for dv in devs), tuple(int(J_idxs[nm][0]) if len(J_idxs[nm]) else 0 for nm in names),
               bool(self.use_g), bool(self.admittance is True), self.nullspace_config is not None)
        hit = self._layouts.get(key)
        if hit is not None:
            return hit
        self._layouts[key] = lay = self._build_layout(devs, J_idxs)
        return lay

    def _build_layout(self, devs, J_idxs) -> OSCLayout:
        return OSCLayout.from_devices(devs, self.robot, use_g=bool(self.use_g), admittance=bool(self.admittance is True),
                                      nullspace=self.nullspace_config is not None, J_idxs=J_idxs)

    def generate(self, targets: Dict[str, Target]):
        if self.robot.is_using_sim() is False:
            assert self.robot.is_running(), "Robot must be running!"
        state = self.robot.get_all_states()
        Js, J_idxs = state[RobotState.J]
        names = list(targets.keys())
        layout = self._layout_for(names, J_idxs)
        ctx = self._ctx.get(layout.key())
        if ctx is None:
            ctx = BatchedOSC(layout, max_batch=1, dtype=self._dtype, hip_device=self._hip_device)
            self._ctx[layout.key()] = ctx

        J = np.vstack([Js[nm] for nm in names])[None]
        M = np.asarray(state[RobotState.M], dtype=np.float64)[None]
        dq = np.asarray(state[RobotState.DQ], dtype=np.float64)[None]
        bias = np.asarray(self.sim.data.qfrc_bias, dtype=np.float64)[self.robot.joint_ids_all][None]
        ee = np.array([[np.concatenate([state[nm][DeviceState.EE_XYZ], state[nm][DeviceState.EE_QUAT]])
                        for nm in names]], dtype=np.float64)
        wr = np.array([[np.concatenate([state[nm][DeviceState.FORCE], state[nm][DeviceState.TORQUE]])
                        for nm in names]], dtype=np.float64)
        tp = np.array([[np.concatenate([targets[nm].get_xyz(), targets[nm].get_quat()])
                        for nm in names]], dtype=np.float64)
        tv = np.array([[np.hstack([targets[nm].get_xyz_vel(), targets[nm].get_abg_vel()])
                        for nm in names]], dtype=np.float64)
        cc = [self.device_configs[nm] for nm in names]
        mv = [self.robot.get_device(nm).max_vel or [0.0, 0.0] for nm in names]
        ctx.set_gains(kp=[c['kp'] for c in cc], kv=[c['kv'] for c in cc], ko=[c['ko'] for c in cc],
                      k=[c['k'] for c in cc], d=[c['d'] for c in cc], max_vel=mv,
                      null_kv=(self.nullspace_config['kv'] if self.nullspace_config is not None else 0.0))
        # one library call per tick: one copy in, the step, one copy out, one synchronisation (irlosc_tick)
        u, fl = ctx.tick(M, J, dq, bias, ee, tp, tgt_vel=tv, wrench=wr, return_flags=True, check_symmetric=True)
        self.last_flags = int(fl[0])
        if self.last_flags & _lib.FLAG_BAD_JIDX:
            raise IndexError("target-velocity branch indexed dx out of range (osc.py:176)")
        u_all = u[0].astype(np.float64)
        forces, force_idxs = [], []
        for nm in names:
            dev = self.robot.sub_devices_dict[nm]
            forces.append(u_all[dev.actuator_trnids])
            force_idxs.append(dev.ctrl_idxs)
        return force_idxs, forces
