"""Host side (no GPU): the build's Device / Robot / MujocoApp assemble the same index maps and
the same M, J, dq, wrench as the reference's classes did on the same simulator state
(fixtures e2e_*.npz hold both the raw sim arrays and the reference's assembled state)."""
import numpy as np
import pytest

import irl_control_amd as ic
from irl_control_amd import DeviceState, RobotState
from conftest import app_from_e2e, load_e2e


@pytest.mark.parametrize("name", ["e2e_gain_test", "e2e_admit_test", "e2e_single_arm"])
def test_assembly_matches_reference(name):
    g = load_e2e(name)
    meta = g["meta"]
    for b in range(g["M"].shape[0]):
        app, robot, osc, targets = app_from_e2e(g, b)
        st = robot.get_all_states()
        Js, J_idxs = st[RobotState.J]
        order = meta["target_order"]
        assert np.array_equal(st[RobotState.M], g["M"][b])
        assert np.array_equal(st[RobotState.DQ], g["dq"][b])
        assert np.array_equal(np.vstack([Js[dn] for dn in order]), g["J"][b])
        wr = np.array([np.concatenate([st[dn][DeviceState.FORCE], st[dn][DeviceState.TORQUE]]) for dn in order])
        assert np.allclose(wr, g["wrench"][b], rtol=0, atol=0)
        for dn in order:
            dev = robot.get_device(dn)
            assert list(dev.joint_ids) == meta["joint_ids"][dn]
            assert list(dev.joint_names) == meta["joint_names"][dn]
        assert {k: list(v) for k, v in J_idxs.items()} == meta["J_idxs"]
        assert [list(robot.get_device(dn).ctrl_idxs) for dn in order] == meta["force_idxs"]


def test_dual_ur5_index_tables():
    """joint / actuator maps of the Dual-UR5."""
    app = ic.MujocoApp("iros2022.yaml", None, sim=ic.FakeSim())
    robot = app.get_robot("DualUR5")
    base, right, left = (robot.get_device(n) for n in ("base", "ur5right", "ur5left"))
    assert list(base.joint_ids_all) == [0]
    assert list(right.joint_ids) == list(range(1, 7)) and list(right.gripper_ids) == list(range(7, 13))
    assert list(left.joint_ids) == list(range(13, 19)) and list(left.gripper_ids) == list(range(19, 25))
    assert list(base.ctrl_idxs) == [0] and list(right.ctrl_idxs) == list(range(1, 8)) \
        and list(left.ctrl_idxs) == list(range(8, 15))
    assert list(right.actuator_trnids) == [1, 2, 3, 4, 5, 6, 10]
    assert list(left.actuator_trnids) == [13, 14, 15, 16, 17, 18, 22]
    assert robot.num_joints_total == 25 and robot.num_scene_joints == 25
    # start angles written into qpos by the constructor (device.py:76-79)
    assert np.allclose(app.sim.data.qpos[0], -1.56)


def test_missing_start_body_reproduces_reference_failure():
    """Without start_body the arm chain swallows the stand joint and the 6 start angles no longer
    fit the 7 joint ids: the reference raises ValueError there; so do we."""
    import yaml, os, tempfile
    from irl_control_amd import mujoco_app
    path = os.path.join(os.path.dirname(mujoco_app.__file__), "robot_configs", "default_xyz.yaml")
    cfg = yaml.safe_load(open(path))
    for d in cfg["devices"]:
        d.pop("start_body", None)
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(cfg, f)
    with pytest.raises(ValueError):
        ic.MujocoApp(f.name, None, sim=ic.FakeSim())
    os.unlink(f.name)


def test_polling_thread_mode_contract():
    sim = ic.FakeSim()
    app = ic.MujocoApp("iros2022.yaml", None, use_sim=False, sim=sim)
    robot = app.get_robot("DualUR5")
    with pytest.raises(AssertionError):
        robot.stop()                      # not running (robot.py:119)
    import threading, time
    th = threading.Thread(target=robot.start)
    th.start()
    time.sleep(0.05)
    assert robot.is_running()
    st = robot.get_all_states()
    assert st[RobotState.M].shape == (25, 25)
    robot.stop()
    th.join(timeout=2)
    assert not th.is_alive()
    app2 = ic.MujocoApp("iros2022.yaml", None, use_sim=True, sim=ic.FakeSim())
    with pytest.raises(AssertionError):
        app2.get_robot("DualUR5").start()  # use_sim must be False (robot.py:104)


def test_target_api():
    t = ic.Target()
    assert np.array_equal(t.get_quat(), [1, 0, 0, 0]) and np.array_equal(t.get_xyz(), np.zeros(3))
    assert np.all(np.hstack([t.get_xyz_vel(), t.get_abg_vel()]) == 0)
    t.set_abg([0.1, -0.2, 0.3])
    assert np.allclose(t.get_abg(), [0.1, -0.2, 0.3])
    with pytest.raises(AssertionError):
        t.set_xyz([1, 2])
    with pytest.raises(AssertionError):
        ic.Target([0] * 5)
    t.set_all_quat([1, 2, 3], [0, 1, 0, 0])
    assert np.array_equal(t.pose7(), [1, 2, 3, 0, 1, 0, 0])


def test_osc_ctor_mutates_gain_dicts_like_reference():
    app = ic.MujocoApp("default_xyz_abg.yaml", None, sim=ic.FakeSim())
    g = app.get_controller_config("osc2")
    ic.OSC(app.get_robot("DualUR5"), app.sim, [("ur5right", g)])
    assert np.array_equal(g["task_space_gains"], [200] * 6)
    assert np.allclose(g["lamb"], np.array([200] * 6) / 50)


def test_full_mass_matrix_dispatches_on_the_sim_object(monkeypatch):
    """the backend is picked by what `sim` is, not by which package imports.  A mujoco_py-style sim must go
    to mujoco_py.cymj._mj_fullM even when a module called `mujoco` is importable, and vice versa."""
    import sys
    import types
    from irl_control_amd import backend
    calls = []
    fake_mujoco = types.ModuleType("mujoco")
    fake_mujoco.mj_fullM = lambda model, dst, qM: (calls.append("mujoco"), dst.__setitem__(slice(None), 2.0))
    fake_py = types.ModuleType("mujoco_py")
    fake_py.cymj = types.SimpleNamespace(_mj_fullM=lambda model, dst, qM: (calls.append("mujoco_py"), dst.__setitem__(slice(None), 3.0)))
    monkeypatch.setitem(sys.modules, "mujoco", fake_mujoco)
    monkeypatch.setitem(sys.modules, "mujoco_py", fake_py)
    PyModel = type("PyMjModel", (), {"nv": 2, "__module__": "mujoco_py.cymj"})
    OffModel = type("MjModel", (), {"nv": 2, "__module__": "mujoco._structs"})
    data = types.SimpleNamespace(qM=np.zeros(3))
    out = np.zeros(4)
    assert backend.backend_of(types.SimpleNamespace(model=PyModel(), data=data)) == "mujoco_py"
    backend.full_mass_matrix(types.SimpleNamespace(model=PyModel(), data=data), out)
    assert calls == ["mujoco_py"] and np.all(out == 3.0)
    backend.full_mass_matrix(types.SimpleNamespace(model=OffModel(), data=data), out)
    assert calls == ["mujoco_py", "mujoco"] and np.all(out == 2.0)
    inj = types.SimpleNamespace(model=PyModel(), data=data, fullM=lambda: np.eye(2))
    backend.full_mass_matrix(inj, out)
    assert calls == ["mujoco_py", "mujoco"] and np.array_equal(out, np.eye(2).reshape(-1))


def test_live_device_mutations_oracle_and_layout_keying():
    """Reference behaviours that only show when a live Device changes between ticks: `ctrlr_dof_abg` switched off / on
    after construction (ps_move_example.py:137-150: the angle error goes to zero, the six task rows stay) and `max_vel[0]`
    re-set before a tick (insertion_task.py:294).  Host side on the CPU: the state assembly + the layout OSC.generate
    would hand to the GPU (calc_abg off, row mask unchanged, one layout object per distinct masking) + the oracle on it,
    against the forces the REFERENCE produced on the same OSC object through the same mutations."""
    import numpy as np
    import irl_control_amd as ic
    from conftest import live_mutation_phases, load_e2e
    from oracle import osc_oracle
    g = load_e2e("e2e_live_mutations")
    meta = g["meta"]
    names = meta["target_order"]
    worst = 0.0
    for b in range(g["qM"].shape[0]):
        layouts = set()
        for p, app, robot, osc, targets in live_mutation_phases(g, b):
            state = robot.get_all_states()
            Js, J_idxs = state[ic.RobotState.J]
            lay = osc._layout_for(names, J_idxs)
            layouts.add(id(lay))
            off = [dn for dn in names if not any(robot.get_device(dn).ctrlr_dof_abg)]
            assert lay.k == 13 and lay.dev_rows == [6, 6, 1]                     # the row mask is the constructor's
            assert [not c for c in lay.calc_abg] == [dn in off for dn in names]
            J = np.vstack([Js[nm] for nm in names])[None]
            ee = np.array([[np.concatenate([state[nm][ic.DeviceState.EE_XYZ], state[nm][ic.DeviceState.EE_QUAT]]) for nm in names]])
            tp = np.array([[np.concatenate([targets[nm].get_xyz(), targets[nm].get_quat()]) for nm in names]])
            cc = [osc.device_configs[nm] for nm in names]
            gains = dict(kp=[c["kp"] for c in cc], kv=[c["kv"] for c in cc], ko=[c["ko"] for c in cc], k=[c["k"] for c in cc],
                         d=[c["d"] for c in cc], max_vel=[robot.get_device(nm).max_vel for nm in names], null_kv=osc.nullspace_config["kv"])
            u = osc_oracle.generate_batch(lay.as_oracle_dict(), gains, state[ic.RobotState.M][None], J, state[ic.RobotState.DQ][None],
                                          app.sim.data.qfrc_bias[robot.joint_ids_all][None], ee, tp)[0]
            flat = np.concatenate([u[robot.get_device(nm).actuator_trnids] for nm in names])
            ref = g["forces_flat"][b, p]
            worst = max(worst, np.max(np.abs(flat - ref)) / np.max(np.abs(ref)))
        assert len(layouts) == 4          # both on / right off / both off / left off: re-keyed, and reused when masks recur
    assert worst <= 1e-9, worst


def test_mujoco_app_says_why_a_relative_scene_cannot_be_found():
    """The reference joins a relative scene_file with its own scenes/ directory (mujoco_app.py:14-16); this package ships no
    scenes, so the error must say so instead of surfacing as a loader failure deep inside a simulator binding."""
    import pytest
    import irl_control_amd as ic
    with pytest.raises(FileNotFoundError, match="ships no scenes/ directory"):
        ic.MujocoApp("default_xyz_abg.yaml", "gain_test_scene.xml")
    with pytest.raises(ValueError, match="sim=.*or scene_file="):
        ic.MujocoApp("default_xyz_abg.yaml", None)
