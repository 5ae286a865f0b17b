"""Headless stand-in for the simulator the OSC path reads its inputs from.

The reference pulls every input of ``OSC.generate`` out of a mujoco_py ``MjSim``
(/root/reference/irl_control/device.py:42-98,125-167, robot.py:26,69, osc.py:191).  MuJoCo is not
available in this image or on the GPU box, so this module provides the *exact* member set those
lines touch, backed by plain numpy arrays:

  model: body_name2id, body_parentid, body_jntadr, body_jntnum, joint_id2name, joint_name2id,
         jnt_qposadr, actuator_trnid, nv, nu
  data : qpos, qvel, qacc, qM, qfrc_bias, sensordata, ctrl, get_body_xpos/xquat/xvelp/jacp/jacr,
         get_site_xmat
  sim  : model, data, forward(), step()

The body / joint / actuator tables below restate the kinematic tree of the Dual-UR5 scene
(/root/reference/irl_control/scenes/world.xml:33-39 + dual_ur5.xml:51-297, include order of
gain_test_scene.xml:5-6): 38 bodies, 25 hinge joints, 15 actuators, 18 sensordata floats.
State arrays are *inputs*: tests and the bench fill them (``randomize``), a dynamics plug-in may
overwrite them in ``forward()``.
"""
from typing import Callable, Dict, List, Optional

import numpy as np

_GRIPPER_LINKS = ["left_outer_knuckle", "left_inner_finger", "left_inner_knuckle",
                  "right_outer_knuckle", "right_inner_finger", "right_inner_knuckle"]
# parent of each gripper link, relative: -1 = the adapter link, else index into _GRIPPER_LINKS
_GRIPPER_PARENT = [-1, 0, -1, -1, 3, -1]


def dual_ur5_tree():
    """(body_names, body_parent, body_joint_names) for world.xml + dual_ur5.xml."""
    names: List[str] = ["world", "target_red", "target_blue", "origin_base", "dual_ur_stand",
                        "ur_stand_dummy"]
    parent: List[int] = [0, 0, 0, 0, 3, 4]
    joints: List[List[str]] = [[], [], [], [], ["ur_stand_joint"], []]
    stand = 4
    for side in ("ur5right", "ur5left"):
        names.append(f"base_link_{side}"); parent.append(stand); joints.append([])
        for li in range(6):
            names.append(f"link{li + 1}_{side}"); parent.append(len(names) - 2)
            joints.append([f"joint{li}_{side}"])
        names.append(f"ur_EE_{side}"); parent.append(len(names) - 2); joints.append([])
        names.append(f"robotiq_85_adapter_link_{side}"); parent.append(len(names) - 2)
        joints.append([])
        adapter = len(names) - 1
        names.append(f"EE_{side}"); parent.append(adapter); joints.append([])
        first = len(names)
        for gi, (gname, gpar) in enumerate(zip(_GRIPPER_LINKS, _GRIPPER_PARENT)):
            names.append(f"{gname}_{side}")
            parent.append(adapter if gpar < 0 else first + gpar)
            joints.append([f"{gname}_joint_{side}"])
    return names, parent, joints


# ctrl index -> joint name (dual_ur5.xml:267-287)
def dual_ur5_actuated_joints():
    out = ["ur_stand_joint"]
    for side in ("ur5right", "ur5left"):
        out += [f"joint{i}_{side}" for i in range(6)] + [f"right_outer_knuckle_joint_{side}"]
    return out


class FakeModel:
    def __init__(self, body_names, body_parent, body_joints, actuated_joints,
                 n_free_bodies: int = 0, free_joint_names=None):
        self.body_names = list(body_names)
        self.body_parentid = np.asarray(body_parent, dtype=np.int32)
        self.joint_names: List[str] = []
        self.body_jntadr = np.full(len(body_names), -1, dtype=np.int32)
        self.body_jntnum = np.zeros(len(body_names), dtype=np.int32)
        for b, js in enumerate(body_joints):
            if js:
                self.body_jntadr[b] = len(self.joint_names)
                self.body_jntnum[b] = len(js)
                self.joint_names += list(js)
        self.nq_robot = len(self.joint_names)
        # optional free bodies appended after the robot (admit/insertion scenes: nv = 25 + 6*m)
        self.n_free_bodies = n_free_bodies
        self.nv = self.nq_robot + 6 * n_free_bodies
        self.nq = self.nq_robot + 7 * n_free_bodies
        # free joints (7 qpos / 6 dofs each) behind the robot's hinges; named so that action objects can be addressed
        # (action_sequence_configs/insertion_task.yaml: joint_name)
        self.free_joint_names = list(free_joint_names) if free_joint_names is not None else \
            [f"free_joint_{i}" for i in range(n_free_bodies)]
        assert len(self.free_joint_names) == n_free_bodies
        self.jnt_qposadr = np.concatenate([np.arange(self.nq_robot), self.nq_robot + 7 * np.arange(n_free_bodies)]).astype(np.int32)
        trn = [self.joint_names.index(j) for j in actuated_joints]
        self.actuator_trnid = np.stack([np.asarray(trn, dtype=np.int32),
                                        np.full(len(trn), -1, dtype=np.int32)], axis=1)
        self.nu = len(trn)
        self.nbody = len(self.body_names)

    def body_name2id(self, name: str) -> int:
        try:
            return self.body_names.index(name)
        except ValueError:
            raise ValueError(f'No "body" with name {name} exists.')

    def joint_id2name(self, jid: int) -> str:
        return self.joint_names[int(jid)]

    def joint_name2id(self, name: str) -> int:
        if name in self.free_joint_names:
            return self.nq_robot + self.free_joint_names.index(name)
        return self.joint_names.index(name)


class FakeData:
    def __init__(self, model: FakeModel, sites: List[str]):
        nb, nv = model.nbody, model.nv
        self._model = model
        self.qpos = np.zeros(model.nq)
        self.qvel = np.zeros(nv)
        self.qacc = np.zeros(nv)
        self.qM = np.eye(nv)              # dense here; real MuJoCo keeps it sparse (see fullM)
        self.qfrc_bias = np.zeros(nv)
        self.sensordata = np.zeros(18)
        self.ctrl = np.zeros(model.nu)
        self.xfrc_applied = np.zeros((nb, 6))
        self.body_xpos = np.zeros((nb, 3))
        self.body_xquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
        self.body_xvelp = np.zeros((nb, 3))
        self.body_jacp = np.zeros((nb, 3 * nv))
        self.body_jacr = np.zeros((nb, 3 * nv))
        self.site_xmat: Dict[str, np.ndarray] = {s: np.eye(3) for s in sites}

    def _bid(self, name): return self._model.body_name2id(name)
    def get_body_xpos(self, name): return self.body_xpos[self._bid(name)]
    def get_body_xquat(self, name): return self.body_xquat[self._bid(name)]
    def get_body_xvelp(self, name): return self.body_xvelp[self._bid(name)]
    def get_body_jacp(self, name): return self.body_jacp[self._bid(name)]
    def get_body_jacr(self, name): return self.body_jacr[self._bid(name)]
    def get_site_xmat(self, name): return self.site_xmat[name]

    def set_mocap_pos(self, name, pos):
        self.body_xpos[self._bid(name)] = pos

    def set_mocap_quat(self, name, quat):
        self.body_xquat[self._bid(name)] = quat

    def get_joint_qpos(self, name):
        """qpos of a joint: one value for a hinge, pos + quat (7) for a free joint (mujoco_py's MjSimState helper)."""
        m = self._model
        adr = m.jnt_qposadr[m.joint_name2id(name)]
        return self.qpos[adr:adr + 7] if name in m.free_joint_names else self.qpos[adr]


class FakeSim:
    """``MjSim`` look-alike.  ``dynamics`` (optional) is called by forward()/step() to refresh the
    derived arrays from qpos/qvel; without it the arrays are whatever the caller wrote."""

    def __init__(self, model: Optional[FakeModel] = None, n_free_bodies: int = 0,
                 dynamics: Optional[Callable[["FakeSim"], None]] = None, free_joint_names=None, mocap_names=()):
        if model is None:
            names, parent, joints = dual_ur5_tree()
            for mn in mocap_names:          # extra mocap bodies of a scene (space_mouse_scene.xml:6-14: plate, hand_ur5right, ...)
                names.append(mn); parent.append(0); joints.append([])
            if free_joint_names is not None:
                n_free_bodies = len(free_joint_names)
            model = FakeModel(names, parent, joints, dual_ur5_actuated_joints(), n_free_bodies, free_joint_names)
        self.model = model
        self.data = FakeData(model, ["ft_frame_ur5right", "ft_frame_ur5left"])
        self.dynamics = dynamics
        self.n_forward = 0
        self.n_step = 0

    def forward(self):
        self.n_forward += 1
        if self.dynamics is not None:
            self.dynamics(self)

    def step(self):
        self.n_step += 1
        if self.dynamics is not None:
            self.dynamics(self, integrate=True)

    def fullM(self) -> np.ndarray:
        """Dense nv x nv inertia matrix (what ``mj_fullM`` expands ``qM`` into)."""
        return np.asarray(self.data.qM).reshape(self.model.nv, self.model.nv)


# ----------------------------------------------------------------------------------------------
# Synthetic state generation: physically plausible magnitudes and sparsity.
# ----------------------------------------------------------------------------------------------
_ARM_COLS = {"ur5right": [0] + list(range(1, 7)), "ur5left": [0] + list(range(13, 19))}
_EE_BODY = {"base": "ur_stand_dummy", "ur5right": "ur_EE_ur5right", "ur5left": "ur_EE_ur5left"}
_GRIPPER_JOINTS = list(range(7, 13)) + list(range(19, 25))


def random_unit_quat(rng, size=None):
    q = rng.normal(size=(4,) if size is None else (size, 4))
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def quat_mul(a, b):
    w1, x1, y1, z1 = np.moveaxis(a, -1, 0)
    w2, x2, y2, z2 = np.moveaxis(b, -1, 0)
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                     w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                     w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2], axis=-1)


def synth_mass_matrix(rng, B: int, nv: int = 25, gripper_scale: float = 1e-2) -> np.ndarray:
    """SPD, exactly symmetric, arm inertias O(1), gripper rows/cols scaled down (tiny links)."""
    A = rng.normal(size=(B, nv, nv))
    M = A @ np.swapaxes(A, 1, 2) / nv
    idx = np.arange(nv)
    M[:, idx, idx] += rng.uniform(0.05, 2.0, size=(B, nv))
    s = np.ones(nv)
    s[[j for j in _GRIPPER_JOINTS if j < nv]] = gripper_scale
    M = M * s[None, :, None] * s[None, None, :]
    return 0.5 * (M + np.swapaxes(M, 1, 2))


def synth_jac6(rng, B: int, device: str, nv: int = 25) -> np.ndarray:
    """[B,6,nv] unmasked EE Jacobian (rows: jacp xyz then jacr xyz) with the tree's sparsity."""
    J = np.zeros((B, 6, nv))
    if device == "base":
        J[:, 5, 0] = 1.0       # yaw about z: only rotational-z row is non-zero
        return J
    cols = _ARM_COLS[device]
    J[:, :3, cols] = rng.normal(0.0, 0.5, size=(B, 3, len(cols)))
    R = rng.normal(size=(B, 3, len(cols)))
    J[:, 3:, cols] = R / np.linalg.norm(R, axis=1, keepdims=True)   # unit rotation axes
    return J


def randomize(sim: FakeSim, rng: np.random.Generator, wrench: bool = False):
    """Fill one FakeSim with a random plausible Dual-UR5 state (single instance)."""
    nv = sim.model.nv
    d = sim.data
    d.qM = synth_mass_matrix(rng, 1, nv)[0]
    d.qvel[:] = rng.normal(0.0, 0.5, size=nv)
    d.qpos[:sim.model.nq_robot] = rng.uniform(-np.pi, np.pi, size=sim.model.nq_robot)
    d.qfrc_bias[:] = rng.normal(0.0, 5.0, size=nv)
    for dev, body in _EE_BODY.items():
        b = sim.model.body_name2id(body)
        J6 = synth_jac6(rng, 1, dev, nv)[0]
        d.body_jacp[b] = J6[:3].reshape(-1)
        d.body_jacr[b] = J6[3:].reshape(-1)
        d.body_xpos[b] = rng.uniform(-1.0, 1.0, size=3)
        d.body_xquat[b] = random_unit_quat(rng)
        d.body_xvelp[b] = rng.normal(0.0, 0.2, size=3)
    if wrench:
        d.sensordata[:] = rng.normal(0.0, 5.0, size=18)
        for s in d.site_xmat:
            q = random_unit_quat(rng)
            from .transforms import quat2mat
            d.site_xmat[s] = quat2mat(q)
    return sim


class ToyDynamics:
    """Deterministic stand-in for physics, for the headless tick loops (examples/headless_loops.py) and their
    per-tick goldens: MuJoCo is in neither this image nor the GPU box.  Nothing here depends on ``ctrl``, so the state
    trajectory is the same whichever controller drives the loop (the reference when the goldens are minted, the HIP
    path in the tests) and the per-tick torques are comparable one to one.  Per ``sim.step()``:

      * every end-effector body with a goal slides ``rate`` of the way towards it (position, and orientation along
        the chord of the two quaternions), so the waypoint logic of gain_test has something to do;
      * Jacobians, joint velocities and bias forces follow slow sinusoids around their initial values;
      * the F/T sensors read the reaction to ``xfrc_applied`` on the listed bodies (the push of admit_test).
    """

    def __init__(self, rate: float = 0.05):
        self.rate = rate
        self.goal_xyz: Dict[str, np.ndarray] = {}
        self.goal_quat: Dict[str, np.ndarray] = {}
        self.ft_bodies = {"left_outer_knuckle_ur5right": 0, "left_outer_knuckle_ur5left": 6}   # body -> sensordata offset
        self.goal_provider = None     # optional callable -> (goal_xyz dict, goal_quat dict), asked at every step
        self.t = 0
        self._init = None

    def __call__(self, sim, integrate: bool = False):
        if not integrate:
            return
        d = sim.data
        if self._init is None:
            self._init = (d.body_jacp.copy(), d.body_jacr.copy(), d.qvel.copy(), d.qfrc_bias.copy(), d.sensordata.copy())
        jp0, jr0, qv0, b0, s0 = self._init
        self.t += 1
        if self.goal_provider is not None:
            self.goal_xyz, self.goal_quat = self.goal_provider()
        for body, xyz in self.goal_xyz.items():
            b = sim.model.body_name2id(body)
            d.body_xpos[b] += self.rate * (np.asarray(xyz, dtype=np.float64) - d.body_xpos[b])
        for body, quat in self.goal_quat.items():
            b = sim.model.body_name2id(body)
            q, g = d.body_xquat[b], np.asarray(quat, dtype=np.float64)
            g = g / np.linalg.norm(g)
            if np.dot(q, g) < 0.0:
                g = -g
            q = q + self.rate * (g - q)
            d.body_xquat[b] = q / np.linalg.norm(q)
        s, c = np.sin(0.07 * self.t), np.cos(0.045 * self.t)
        d.body_jacp[:] = jp0 * (1.0 + 0.05 * s)
        d.body_jacr[:] = jr0 * (1.0 + 0.03 * c)
        d.qvel[:] = qv0 * (0.98 ** self.t) + 0.05 * s
        d.qfrc_bias[:] = b0 * (1.0 + 0.1 * c)
        d.sensordata[:] = s0
        for body, off in self.ft_bodies.items():
            b = sim.model.body_name2id(body)
            d.sensordata[off:off + 6] = s0[off:off + 6] - d.xfrc_applied[b]
