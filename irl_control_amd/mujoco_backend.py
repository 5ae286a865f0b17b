"""``MujocoSim``: the official ``mujoco`` bindings behind the simulator protocol this package reads
(SURVEY.md section 8 row f2).

The reference is written against mujoco_py (``MjSim``: /root/reference/irl_control/mujoco_app.py:17-18) while its
README points users to the official bindings.  Everything Device / Robot / OSC / the examples touch is the member set
listed in fakesim.py; this adapter provides exactly that set on top of ``mujoco.MjModel`` / ``mujoco.MjData``:

  model: body_name2id, body_parentid, body_jntadr, body_jntnum, joint_id2name, joint_name2id, jnt_qposadr,
         actuator_trnid, nv, nu
  data : qpos, qvel, qacc, qM, qfrc_bias, sensordata, ctrl, xfrc_applied, get_body_xpos / xquat / xvelp / jacp / jacr,
         get_site_xmat, set_mocap_pos / set_mocap_quat, get_joint_qpos / get_joint_qvel / set_joint_qpos / set_joint_qvel
  sim  : model, data, forward(), step(), fullM(), inverse()

``import mujoco`` happens only when an adapter is created, so the package works without MuJoCo (FakeSim, or the GPU
front end of rigid_body.py, which needs no simulator at all).  Neither MuJoCo package exists in the build image or on
the GPU box: tests/test_mujoco_backend.py drives this file against a stand-in module with the same call signatures.
"""
import numpy as np


class _Model:
    def __init__(self, mj, m):
        self._mj, self._m = mj, m

    def __getattr__(self, name):                    # nv, nu, body_parentid, body_jntadr, jnt_qposadr, actuator_trnid, ...
        return getattr(self._m, name)

    def _id(self, kind, name, what):
        i = self._mj.mj_name2id(self._m, kind, name)
        if i < 0:
            raise ValueError(f'No "{what}" with name {name} exists.')       # mujoco_py's message (device.py:42-58 relies on it)
        return i

    def body_name2id(self, name):
        return self._id(self._mj.mjtObj.mjOBJ_BODY, name, "body")

    def joint_name2id(self, name):
        return self._id(self._mj.mjtObj.mjOBJ_JOINT, name, "joint")

    def site_name2id(self, name):
        return self._id(self._mj.mjtObj.mjOBJ_SITE, name, "site")

    def joint_id2name(self, jid):
        return self._mj.mj_id2name(self._m, self._mj.mjtObj.mjOBJ_JOINT, int(jid))


class _Data:
    def __init__(self, mj, model: _Model, d):
        self._mj, self._model, self._m, self._d = mj, model, model._m, d

    def __getattr__(self, name):                    # qpos, qvel, qacc, qM, qfrc_bias, sensordata, ctrl, xfrc_applied, ...
        return getattr(self._d, name)

    def get_body_xpos(self, name):
        return self._d.xpos[self._model.body_name2id(name)]

    def get_body_xquat(self, name):
        return self._d.xquat[self._model.body_name2id(name)]

    def get_body_xvelp(self, name):
        v = np.zeros(6)
        self._mj.mj_objectVelocity(self._m, self._d, self._mj.mjtObj.mjOBJ_BODY, self._model.body_name2id(name), v, 0)
        return v[3:]                                 # (angular, linear) in world orientation

    def _jac(self, name):
        nv = self._m.nv
        jp, jr = np.zeros((3, nv)), np.zeros((3, nv))
        self._mj.mj_jacBody(self._m, self._d, jp, jr, self._model.body_name2id(name))
        return jp, jr

    def get_body_jacp(self, name):
        return self._jac(name)[0].reshape(-1)        # flat 3*nv like mujoco_py (device.py:125 reshapes it)

    def get_body_jacr(self, name):
        return self._jac(name)[1].reshape(-1)

    def get_site_xmat(self, name):
        return np.asarray(self._d.site_xmat[self._model.site_name2id(name)]).reshape(3, 3)

    def set_mocap_pos(self, name, pos):
        self._d.mocap_pos[self._m.body_mocapid[self._model.body_name2id(name)]] = pos

    def set_mocap_quat(self, name, quat):
        self._d.mocap_quat[self._m.body_mocapid[self._model.body_name2id(name)]] = quat

    # mujoco_py's per-joint accessors (examples/insertion_task.py:227,251 read the pose of an action object's free joint):
    # width by joint type -- mjJNT_FREE 7 qpos / 6 dofs, mjJNT_BALL 4 / 3, slide and hinge 1 / 1 (returned as a scalar).
    _QPOS_WIDTH = {0: 7, 1: 4, 2: 1, 3: 1}
    _QVEL_WIDTH = {0: 6, 1: 3, 2: 1, 3: 1}

    def _joint_span(self, name, vel):
        j = self._model.joint_name2id(name)
        adr = int((self._m.jnt_dofadr if vel else self._m.jnt_qposadr)[j])
        return adr, (self._QVEL_WIDTH if vel else self._QPOS_WIDTH)[int(self._m.jnt_type[j])]

    def get_joint_qpos(self, name):
        adr, w = self._joint_span(name, False)
        return self._d.qpos[adr] if w == 1 else self._d.qpos[adr:adr + w]

    def get_joint_qvel(self, name):
        adr, w = self._joint_span(name, True)
        return self._d.qvel[adr] if w == 1 else self._d.qvel[adr:adr + w]

    def set_joint_qpos(self, name, value):
        adr, w = self._joint_span(name, False)
        self._d.qpos[adr:adr + w] = value

    def set_joint_qvel(self, name, value):
        adr, w = self._joint_span(name, True)
        self._d.qvel[adr:adr + w] = value


class MujocoSim:
    """``MjSim`` look-alike over the official bindings."""

    def __init__(self, model, data=None):
        import mujoco
        self._mj = mujoco
        self.mj_model = model
        self.mj_data = data if data is not None else mujoco.MjData(model)
        self.model = _Model(mujoco, model)
        self.data = _Data(mujoco, self.model, self.mj_data)

    @classmethod
    def from_xml_path(cls, path: str) -> "MujocoSim":
        import mujoco
        return cls(mujoco.MjModel.from_xml_path(path))

    def forward(self):
        self._mj.mj_forward(self.mj_model, self.mj_data)

    def step(self):
        self._mj.mj_step(self.mj_model, self.mj_data)

    def inverse(self):
        """mj_inverse (examples/force_test.py:113 calls it after each step to refresh the F/T readings)."""
        self._mj.mj_inverse(self.mj_model, self.mj_data)

    def fullM(self) -> np.ndarray:
        nv = self.mj_model.nv
        out = np.zeros((nv, nv))
        self._mj.mj_fullM(self.mj_model, out, self.mj_data.qM)
        return out
