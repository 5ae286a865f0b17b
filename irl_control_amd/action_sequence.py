"""``ActionSequenceRunner``: the WP / GRIP action-sequence state machine of the reference's insertion demo
(/root/reference/irl_control/examples/insertion_task.py:105-318, action list and action objects in
action_sequence_configs/insertion_task.yaml:35-103), headless and with the simulator injected.

Per tick it is the same hot path as every other caller — ``controller.generate(targets)``, forces into ``sim.data.ctrl``,
``sim.step()`` — around which the demo adds:
  * WP   go to a waypoint: the target is a list, 'start_pos', or an ACTION OBJECT (its free-joint pose in the simulator
         plus a named offset; the orientation target composes the object's yaw with the default gripper orientation,
         insertion_task.py:249-262); the arm's ``max_vel[0]`` is re-set every tick from the current error,
         clip(kp * error, min_speed_xyz, max_speed_xyz) (insertion_task.py:291-295), until the error drops under max_error;
  * GRIP hold the targets and drive the gripper actuator for a duration (wall-clock seconds in the reference,
         insertion_task.py:193-205; ``gripper_duration / tick_seconds`` ticks here, so that a run is reproducible).
``controller`` may be this package's OSC (HIP path) or anything with generate / calc_error (the golden of
tests/golden/loop_insertion_wp.npz was minted by the reference's own InsertionTask methods on the same FakeSim).
"""
import copy
import os
from enum import Enum
from typing import Dict, List, Optional

import numpy as np
import yaml

from .device import DeviceState
from .targets import Target
from .transforms import compose, euler2mat, euler2quat, mat2euler, quat2mat

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
DEFAULT_EE_ROT = np.deg2rad([0, -90, -90])                       # insertion_task.py:18
DEFAULT_EE_QUAT = euler2quat(*DEFAULT_EE_ROT)                    # insertion_task.py:20
GRIPPER_CTRL_IDX = {"ur5right": 7, "ur5left": 14}                # insertion_task.py:152-155


class Action(Enum):
    WP = 0
    GRIP = 1


DEFAULT_PARAMS = {                                               # insertion_task.py:83-103
    Action.WP: dict(kp=6, max_error=0.0018, gripper_force=0.0, min_speed_xyz=0.1, max_speed_xyz=3.0),
    Action.GRIP: dict(gripper_force=-0.08, gripper_duration=1.0),
}


def load_action_config(config_file: str = "insertion_task.yaml") -> Dict:
    path = config_file if os.path.isabs(config_file) else os.path.join(_PKG_DIR, "action_sequence_configs", config_file)
    with open(path) as f:
        return yaml.safe_load(f)


class ActionSequenceRunner:
    def __init__(self, app, controller, active_arm: str = "right", tick_seconds: float = 0.001, max_ticks_per_action: int = 20000,
                 on_tick=None):
        self.app, self.sim, self.controller = app, app.sim, controller
        self.robot = app.get_robot("DualUR5")
        self.ur5right, self.ur5left = self.robot.get_device("ur5right"), self.robot.get_device("ur5left")
        self.set_active_arm(active_arm)
        self.errors: Dict[str, float] = {}
        self.targets = {self.active_arm.name: Target(), self.passive_arm.name: Target()}
        self.action_objects: Dict = {}
        self.start_pos = None
        self.tick_seconds = tick_seconds
        self.max_ticks_per_action = max_ticks_per_action
        self.on_tick = on_tick                 # callback(runner, ctrlr_output) after every simulator step
        self.ticks = 0

    # ---- configuration ----------------------------------------------------------------------------------------------
    def set_active_arm(self, active_arm: str):
        assert active_arm in ("right", "left"), "Demo only supports Dual UR5 configuration"
        self.active_arm, self.passive_arm = (self.ur5right, self.ur5left) if active_arm == "right" else (self.ur5left, self.ur5right)

    def initialize_action_objects(self):
        """Place the action objects (their free joints) as the action-object table says (insertion_task.py:299-311)."""
        for obj in self.action_objects.values():
            quat = euler2quat(*np.deg2rad(obj["initial_pos_abg"])) if "initial_pos_abg" in obj else None
            self.app.set_free_joint_qpos(obj["joint_name"], quat=quat, pos=obj.get("initial_pos_xyz"))

    # ---- per-tick plumbing -------------------------------------------------------------------------------------------
    def send_forces(self, forces, gripper_force: Optional[float] = None, update_errors=None):
        for force_idx, force in zip(*forces):
            self.sim.data.ctrl[force_idx] = force
        if gripper_force:
            self.sim.data.ctrl[GRIPPER_CTRL_IDX[self.active_arm.name]] = gripper_force
        self.sim.step()
        self.ticks += 1
        names = [update_errors] if isinstance(update_errors, str) else (update_errors or [])
        for name in names:
            self.errors[name] = float(np.linalg.norm(self.controller.calc_error(self.targets[name], self.robot.get_device(name))))
        if self.on_tick is not None:
            self.on_tick(self, forces)

    # ---- actions ---------------------------------------------------------------------------------------------------------
    def set_waypoint_targets(self, params: Dict):
        self.targets[self.passive_arm.name].set_xyz(self.passive_arm.get_state(DeviceState.EE_XYZ))
        self.targets[self.passive_arm.name].set_quat(DEFAULT_EE_QUAT)
        if "target_xyz" not in params:
            raise KeyError("target_xyz")
        offset = params.get("offset", [0.0, 0.0, 0.0])
        txyz = params["target_xyz"]
        if isinstance(txyz, str):
            if txyz == "start_pos":
                target = self.start_pos
            else:
                obj = self.action_objects[txyz]
                if isinstance(offset, str):
                    offset = obj[offset]
                target = self.sim.data.get_joint_qpos(obj["joint_name"])[:3] + offset
        elif isinstance(txyz, list):
            target = txyz + offset                                   # list concatenation, as in the reference (:240)
        else:
            raise ValueError("Invalid type for target_xyz!")
        self.targets[self.active_arm.name].set_xyz(target)
        if "target_abg" in params:
            tabg = params["target_abg"]
            if isinstance(tabg, str):
                obj = self.action_objects[tabg]
                obj_quat = self.sim.data.get_joint_qpos(obj["joint_name"])[-4:]
                grip_eul = DEFAULT_EE_ROT + [0, 0, np.deg2rad(obj["grip_yaw"])]
                tfmat = np.matmul(compose([0, 0, 0], quat2mat(obj_quat), [1, 1, 1]),
                                  compose([0, 0, 0], euler2mat(*grip_eul), [1, 1, 1]))
                target_abg = np.array(mat2euler(tfmat[:3, :3]))
            elif isinstance(tabg, list):
                target_abg = np.deg2rad(tabg)
            else:
                raise ValueError("Invalid type for target_abg!")
            self.targets[self.active_arm.name].set_abg(target_abg)
        else:
            self.targets[self.active_arm.name].set_quat(DEFAULT_EE_QUAT)

    @staticmethod
    def _with_defaults(params: Dict, action: Action) -> Dict:
        for key, val in DEFAULT_PARAMS[action].items():
            params.setdefault(key, val)
        return params

    def go_to_waypoint(self, params: Dict):
        assert params["action"] == "WP"
        self.set_waypoint_targets(params)
        self._with_defaults(params, Action.WP)
        arm = self.active_arm.name
        self.errors[arm] = np.inf
        n = 0
        while self.errors[arm] > params["max_error"]:
            # error-adaptive velocity limit of the active arm (insertion_task.py:293-295)
            self.active_arm.max_vel[0] = max(params["min_speed_xyz"], min(params["max_speed_xyz"], params["kp"] * self.errors[arm]))
            self.send_forces(self.controller.generate(self.targets), gripper_force=params["gripper_force"], update_errors=arm)
            n += 1
            if n >= self.max_ticks_per_action:
                raise RuntimeError(f"waypoint not reached within {n} ticks (error {self.errors[arm]:.4g})")

    def grip(self, params: Dict):
        assert params["action"] == "GRIP"
        self._with_defaults(params, Action.GRIP)
        for _ in range(max(1, int(round(params["gripper_duration"] / self.tick_seconds)))):
            self.send_forces(self.controller.generate(self.targets), gripper_force=params["gripper_force"],
                             update_errors=self.active_arm.name)

    def run_sequence(self, action_sequence: List[Dict]):
        self.start_pos = np.copy(self.active_arm.get_state(DeviceState.EE_XYZ))
        for entry in action_sequence:
            entry = copy.deepcopy(entry)
            {"WP": self.go_to_waypoint, "GRIP": self.grip}[entry["action"]](entry)


# ---- the same state machine for a fleet: B robots, B independent object placements, one GPU step per tick --------------
def _calc_error_batch(ee_pose, tgt_pose):
    """OSC.calc_error (osc.py:101-118) for B poses: [xyz error, sxyz Euler angles of q_ee * conj(normalised q_tgt)]."""
    from .transforms import normalized_vector, qconjugate, qmult, quat2euler
    e = np.zeros((len(ee_pose), 6))
    e[:, :3] = ee_pose[:, :3] - tgt_pose[:, :3]
    for b in range(len(ee_pose)):
        q_r = np.array(qmult(normalized_vector(tgt_pose[b, 3:]), qconjugate(ee_pose[b, 3:])))
        e[b, 3:] = quat2euler(qconjugate(q_r))
    return e


class FleetActionSequenceRunner:
    """B robots run the SAME WP / GRIP action list, each on its own action objects (randomised poses), in lockstep on the
    batched controller: per tick one `BatchedOSC` step from joint coordinates (rigid-body front end on the GPU) with
    PER-INSTANCE gains, because the error-adaptive velocity limit of insertion_task.py:293-295 differs per robot.
    An instance that has finished its list holds its last targets.  The physics is whatever `integrate(q, qd, u, rec)`
    does (examples/insertion_fleet_headless.py: M qacc = u - bias with M, bias read back from HBM).

    Semantics per instance are those of ActionSequenceRunner (and the kernels make an instance's result independent of
    its batch-mates), so robot b of a fleet follows, bit for bit, the trajectory it follows alone
    (tests/test_gpu_parity.py::test_fleet_action_sequence_lockstep_equals_solo_runs)."""

    def __init__(self, osc, base_gains: Dict, objects: List[Dict], sequence: List[Dict], active_arm: str = "right",
                 tick_seconds: float = 0.001, passive_hold_orientation: bool = False, integrator_records=()):
        self.osc, self.lay = osc, osc.layout
        # records tick() hands back besides the EE poses it needs itself: a host-side integrator names what IT reads (e.g.
        # ("M", "bias")); by default only the 168 B of EE poses per robot cross PCIe per tick, not the 8.5 KB of records
        self.records = ("ee_pose",) + tuple(k for k in integrator_records if k != "ee_pose")
        # the reference sends the passive arm to DEFAULT_EE_QUAT at every waypoint (insertion_task.py:213), which suits the
        # start pose of its scene; with arbitrary start poses holding the current orientation keeps that arm where it is
        self.passive_hold_orientation = passive_hold_orientation
        self.B = len(objects)
        self.objects = objects                      # per instance: {name: {"pos": xyz, "quat": wxyz, **offsets, "grip_yaw"}}
        self.seq = [copy.deepcopy(e) for e in sequence]
        for e in self.seq:
            ActionSequenceRunner._with_defaults(e, Action.WP if e["action"] == "WP" else Action.GRIP)
        self.arm = "ur5right" if active_arm == "right" else "ur5left"
        self.other = "ur5left" if active_arm == "right" else "ur5right"
        self.ia, self.io = self.lay.dev_names.index(self.arm), self.lay.dev_names.index(self.other)
        self.base_gains = base_gains
        self.tick_seconds = tick_seconds
        self.action = np.zeros(self.B, dtype=int)                 # index of the action each instance is in
        self.grip_left = np.zeros(self.B, dtype=int)              # ticks left in a GRIP action
        self.err = np.full(self.B, np.inf)
        self.max_vel0 = np.zeros(self.B)
        self.gripper_force = np.zeros(self.B)
        self.tgt = None
        self.start_pos = None
        self.entered = np.full(self.B, -1)                        # action whose targets are currently set
        self.ticks = 0

    def _waypoint_target(self, b, params, ee_pose_b):
        obj = self.objects[b]
        offset = params.get("offset", [0.0, 0.0, 0.0])
        txyz = params["target_xyz"]
        if isinstance(txyz, str):
            if txyz == "start_pos":
                xyz = self.start_pos[b]
            else:
                o = obj[txyz]
                xyz = np.asarray(o["pos"]) + np.asarray(o[offset] if isinstance(offset, str) else offset)
        else:
            xyz = np.asarray(txyz, dtype=np.float64) + np.asarray(offset, dtype=np.float64)
        if "target_abg" in params:
            tabg = params["target_abg"]
            if isinstance(tabg, str):
                o = obj[tabg]
                grip_eul = DEFAULT_EE_ROT + [0, 0, np.deg2rad(o["grip_yaw"])]
                R = quat2mat(o["quat"]) @ euler2mat(*grip_eul)
                quat = euler2quat(*mat2euler(R))
            else:
                quat = euler2quat(*np.deg2rad(tabg))
        else:
            quat = DEFAULT_EE_QUAT
        return np.concatenate([xyz, quat])

    def done(self):
        return self.action >= len(self.seq)

    def tick(self, q, qd):
        """One control tick for the fleet: returns (u[B,n], records) for the integrator (records: `integrator_records`)."""
        osc = self.osc
        osc.upload_q(q, qd)
        osc.frontend()
        rec = osc.download_records(keys=self.records)
        ee = rec["ee_pose"].astype(np.float64)
        if self.tgt is None:
            self.tgt = ee.copy()
            self.start_pos = ee[:, self.ia, :3].copy()
        for b in range(self.B):
            a = self.action[b]
            if a >= len(self.seq) or self.entered[b] == a:
                continue
            p = self.seq[a]
            self.entered[b] = a
            self.gripper_force[b] = p["gripper_force"]
            if p["action"] == "WP":
                self.tgt[b, self.io, :3] = ee[b, self.io, :3]            # passive arm holds position (insertion_task.py:212-213)
                self.tgt[b, self.io, 3:] = ee[b, self.io, 3:] if self.passive_hold_orientation else DEFAULT_EE_QUAT
                self.tgt[b, self.ia] = self._waypoint_target(b, p, ee[b])
                self.err[b] = np.inf
            else:
                self.grip_left[b] = max(1, int(round(p["gripper_duration"] / self.tick_seconds)))
        # error-adaptive velocity limit of the active arm, per instance
        mv = np.broadcast_to(np.asarray(self.base_gains["max_vel"], dtype=np.float64), (self.B, self.lay.ndev, 2)).copy()
        for b in range(self.B):
            a = self.action[b]
            if a < len(self.seq) and self.seq[a]["action"] == "WP":
                p = self.seq[a]
                self.max_vel0[b] = max(p["min_speed_xyz"], min(p["max_speed_xyz"], p["kp"] * self.err[b]))
            mv[b, self.ia, 0] = self.max_vel0[b] if self.max_vel0[b] > 0 else mv[b, self.ia, 0]
        g = self.base_gains
        osc.set_gains(g["kp"], g["kv"], g["ko"], g["k"], g["d"], mv, np.full(self.B, g["null_kv"]) if np.ndim(g["null_kv"]) == 0 else g["null_kv"])
        osc.set_targets(self.tgt)
        u = osc.step()
        self.ticks += 1
        return u, rec

    def after_step(self, ee_pose):
        """Bookkeeping after the simulator step (errors are measured on the new state, as in send_forces)."""
        e = np.linalg.norm(_calc_error_batch(ee_pose[:, self.ia], self.tgt[:, self.ia]), axis=1)
        for b in range(self.B):
            a = self.action[b]
            if a >= len(self.seq):
                continue
            self.err[b] = e[b]
            if self.seq[a]["action"] == "WP":
                if self.err[b] <= self.seq[a]["max_error"]:
                    self.action[b] += 1
            else:
                self.grip_left[b] -= 1
                if self.grip_left[b] <= 0:
                    self.action[b] += 1
