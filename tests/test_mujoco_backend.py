"""mujoco_backend.MujocoSim against a stand-in ``mujoco`` module with the official
bindings' call signatures (MuJoCo itself is in neither image): every member Device / Robot / OSC read must come out in
mujoco_py's shapes, and the whole host assembly must run on it and equal the same state fed through FakeSim."""
import sys
import types

import numpy as np
import pytest

from irl_control_amd import fakesim


def _fake_mujoco(fs: "fakesim.FakeSim"):
    """A module object that answers the official API from the arrays of a FakeSim."""
    mj = types.ModuleType("mujoco")
    mj.mjtObj = types.SimpleNamespace(mjOBJ_BODY=1, mjOBJ_JOINT=3, mjOBJ_SITE=6)
    sites = list(fs.data.site_xmat)

    class MjModel:
        __module__ = "mujoco._structs"
        nv, nu = fs.model.nv, fs.model.nu
        body_parentid, body_jntadr, body_jntnum = fs.model.body_parentid, fs.model.body_jntadr, fs.model.body_jntnum
        jnt_qposadr, actuator_trnid = fs.model.jnt_qposadr, fs.model.actuator_trnid
        jnt_type = np.array([3] * fs.model.nq_robot + [0] * fs.model.n_free_bodies)          # mjJNT_HINGE, mjJNT_FREE
        jnt_dofadr = np.concatenate([np.arange(fs.model.nq_robot), fs.model.nq_robot + 6 * np.arange(fs.model.n_free_bodies)])
        body_mocapid = np.array([0 if n == "target_red" else 1 if n == "target_blue" else -1 for n in fs.model.body_names])

    class MjData:
        def __init__(self, m):
            d = fs.data
            self.qpos, self.qvel, self.qacc, self.qM = d.qpos, d.qvel, d.qacc, d.qM
            self.qfrc_bias, self.sensordata, self.ctrl, self.xfrc_applied = d.qfrc_bias, d.sensordata, d.ctrl, d.xfrc_applied
            self.xpos, self.xquat = d.body_xpos, d.body_xquat
            self.site_xmat = np.array([d.site_xmat[s].reshape(-1) for s in sites])
            self.mocap_pos = np.zeros((2, 3))

    def name2id(m, kind, name):
        table = {1: fs.model.body_names, 3: fs.model.joint_names + fs.model.free_joint_names, 6: sites}[kind]
        return table.index(name) if name in table else -1

    def jac(m, d, jp, jr, bid):
        jp[:] = fs.data.body_jacp[bid].reshape(3, -1)
        jr[:] = fs.data.body_jacr[bid].reshape(3, -1)

    def objvel(m, d, kind, bid, out, local):
        out[3:] = fs.data.body_xvelp[bid]

    def fullM(m, dst, qM):
        dst[:] = np.asarray(qM).reshape(dst.shape)

    calls = []
    mj.MjModel, mj.MjData = MjModel, MjData
    mj.mj_name2id, mj.mj_id2name = name2id, lambda m, kind, i: fs.model.joint_names[i]
    mj.mj_jacBody, mj.mj_objectVelocity, mj.mj_fullM = jac, objvel, fullM
    mj.mj_forward = lambda m, d: calls.append("forward")
    mj.mj_step = lambda m, d: calls.append("step")
    mj.mj_inverse = lambda m, d: calls.append("inverse")
    mj._calls = calls
    return mj


def test_adapter_serves_the_member_set_and_the_host_assembly(monkeypatch):
    import irl_control_amd as ic
    from irl_control_amd import backend
    from irl_control_amd.mujoco_backend import MujocoSim
    fs = fakesim.randomize(fakesim.FakeSim(), np.random.default_rng(3), wrench=True)
    mj = _fake_mujoco(fs)
    monkeypatch.setitem(sys.modules, "mujoco", mj)
    sim = MujocoSim(mj.MjModel())
    assert backend.backend_of(sim) == "injected"
    assert sim.model.body_name2id("ur_EE_ur5left") == fs.model.body_name2id("ur_EE_ur5left")
    with pytest.raises(ValueError, match='No "body" with name nope exists'):
        sim.model.body_name2id("nope")
    assert sim.data.get_body_jacp("ur_EE_ur5right").shape == (3 * fs.model.nv,)
    assert np.array_equal(sim.data.get_site_xmat("ft_frame_ur5left"), fs.data.site_xmat["ft_frame_ur5left"])
    assert np.array_equal(sim.fullM(), fs.fullM())
    sim.forward(); sim.step(); sim.inverse()
    assert mj._calls == ["forward", "step", "inverse"]
    sim.data.set_mocap_pos("target_blue", [1, 2, 3])
    assert np.array_equal(sim.mj_data.mocap_pos[1], [1, 2, 3])
    # the whole state assembly (Device / Robot) on the adapter equals the same on FakeSim, bit for bit
    app_a = ic.MujocoApp("default_xyz_abg.yaml", None, sim=sim)
    app_f = ic.MujocoApp("default_xyz_abg.yaml", None, sim=fs)
    sa, sf = app_a.get_robot("DualUR5").get_all_states(), app_f.get_robot("DualUR5").get_all_states()
    assert np.array_equal(sa[ic.RobotState.M], sf[ic.RobotState.M]) and np.array_equal(sa[ic.RobotState.DQ], sf[ic.RobotState.DQ])
    for nm in ("base", "ur5right", "ur5left"):
        assert np.array_equal(sa[ic.RobotState.J][0][nm], sf[ic.RobotState.J][0][nm])
        for key in (ic.DeviceState.EE_XYZ, ic.DeviceState.EE_QUAT, ic.DeviceState.FORCE, ic.DeviceState.TORQUE):
            assert np.array_equal(sa[nm][key], sf[nm][key])


def test_action_sequence_runner_reads_object_poses_through_the_adapter(monkeypatch):
    """ActionSequenceRunner.set_waypoint_targets reads an action object's free-joint pose with sim.data.get_joint_qpos
    (insertion_task.py:227,251): with the official bindings that accessor is the adapter's, not MjData's."""
    import irl_control_amd as ic
    from irl_control_amd.action_sequence import ActionSequenceRunner, load_action_config
    from irl_control_amd.mujoco_backend import MujocoSim
    free = ["free_joint_grommet_11mm", "free_joint_dual_peg", "free_joint_female", "free_joint_male"]
    fs = fakesim.randomize(fakesim.FakeSim(free_joint_names=free), np.random.default_rng(5))
    mj = _fake_mujoco(fs)
    monkeypatch.setitem(sys.modules, "mujoco", mj)
    sim = MujocoSim(mj.MjModel())
    app = ic.MujocoApp("default_xyz_abg.yaml", None, sim=sim)

    class NoController:
        pass
    r = ActionSequenceRunner(app, NoController(), active_arm="left")
    cfg = load_action_config()
    r.action_objects = cfg["grommet_action_objects"]
    r.initialize_action_objects()                                    # MujocoApp.set_free_joint_qpos on the adapter
    male = cfg["grommet_action_objects"]["male_object"]
    got = sim.data.get_joint_qpos(male["joint_name"])
    assert got.shape == (7,) and np.allclose(got[:3], male["initial_pos_xyz"])
    assert np.array_equal(got, fs.data.get_joint_qpos(male["joint_name"]))
    assert np.isscalar(sim.data.get_joint_qpos("joint0_ur5right")) or np.ndim(sim.data.get_joint_qpos("joint0_ur5right")) == 0
    assert np.shape(sim.data.get_joint_qvel(male["joint_name"])) == (6,)
    sim.data.set_joint_qvel(male["joint_name"], np.arange(6.0))
    dof0 = fs.model.nq_robot + 6 * free.index(male["joint_name"])
    assert np.array_equal(fs.data.qvel[dof0:dof0 + 6], np.arange(6.0))
    sim.data.set_joint_qpos("joint1_ur5left", 0.25)
    assert fs.data.qpos[fs.model.joint_names.index("joint1_ur5left")] == 0.25
    r.set_waypoint_targets(dict(action="WP", target_xyz="male_object", target_abg="male_object", offset="hover_offset"))
    assert np.allclose(r.targets["ur5left"].get_xyz(), np.array(male["initial_pos_xyz"]) + male["hover_offset"])


def test_mujocosim_on_the_real_mujoco_reproduces_the_fixture():
    """Row f2 against a REAL `mujoco` (mujoco_app.py:17-18 of the reference loads a real model): MujocoSim on the reference's scene,
    set to the states of tests/golden/mj_dual_ur5.npz, must hand Device / Robot the very arrays the fixture recorded (fullM,
    jacp / jacr, qfrc_bias, xpos / xquat, site_xmat).  Needs both the package and the fixture (oracle/make_mujoco_golden.py);
    SKIPs with the reason otherwise -- neither image has MuJoCo."""
    import os
    mujoco = pytest.importorskip("mujoco", reason="PARITY WITH MUJOCO UNPINNED: the official `mujoco` package is in neither the build "
                                                  "image nor the GPU box (MujocoSim is exercised on a stand-in module above)")
    if not hasattr(mujoco, "MjModel") or not hasattr(mujoco.MjModel, "from_xml_path"):
        pytest.skip("a stand-in `mujoco` module is installed in sys.modules, not the real package")
    fx = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mj_dual_ur5.npz")
    scene = os.environ.get("IRLOSC_MUJOCO_SCENE", "/root/reference/irl_control/scenes/gain_test_scene.xml")
    if not os.path.exists(fx) or not os.path.exists(scene):
        pytest.skip("tests/golden/mj_dual_ur5.npz or the reference's scene (IRLOSC_MUJOCO_SCENE) absent: run oracle/make_mujoco_golden.py")
    from irl_control_amd.mujoco_backend import MujocoSim
    z = np.load(fx)
    sim = MujocoSim.from_xml_path(scene)
    for i in range(min(8, z["qpos"].shape[0])):
        sim.data.qpos[:] = z["qpos"][i]
        sim.data.qvel[:] = z["qvel"][i]
        sim.forward()
        nv = z["fullM"].shape[1]
        assert np.allclose(sim.fullM().reshape(nv, nv), z["fullM"][i], rtol=0, atol=1e-12)
        assert np.allclose(sim.data.qfrc_bias, z["qfrc_bias"][i], rtol=0, atol=1e-12)
        for e, nm in enumerate(z["ee_bodies"]):
            assert np.allclose(np.asarray(sim.data.get_body_jacp(str(nm))).reshape(3, nv), z["jacp"][i, e], atol=1e-12)
            assert np.allclose(np.asarray(sim.data.get_body_jacr(str(nm))).reshape(3, nv), z["jacr"][i, e], atol=1e-12)
            assert np.allclose(sim.data.get_body_xpos(str(nm)), z["xpos"][i, e], atol=1e-12)
            assert np.allclose(sim.data.get_body_xquat(str(nm)), z["xquat"][i, e], atol=1e-12)
        for s_, nm in enumerate(z["ft_sites"]):
            assert np.allclose(np.asarray(sim.data.get_site_xmat(str(nm))).reshape(9), z["site_xmat"][i, s_], atol=1e-12)
