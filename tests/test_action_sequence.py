"""ActionSequenceRunner on the CPU: the state machine around the hot path, with a
stand-in controller (the HIP-path run against the reference-minted golden is tests/test_gpu_parity.py)."""
import numpy as np

import irl_control_amd as ic
from irl_control_amd.action_sequence import (DEFAULT_EE_QUAT, GRIPPER_CTRL_IDX, ActionSequenceRunner, load_action_config)
from irl_control_amd.fakesim import FakeSim, ToyDynamics, randomize
from irl_control_amd.device import DeviceState

FREE = ["free_joint_grommet_11mm", "free_joint_dual_peg", "free_joint_female", "free_joint_male"]


class ZeroController:
    """generate() -> zero forces with the reference's output structure; calc_error like OSC.calc_error."""

    def __init__(self, robot, real):
        self.robot, self.real, self.calls = robot, real, 0

    def generate(self, targets):
        self.calls += 1
        devs = [self.robot.get_device(n) for n in targets]
        return [d.ctrl_idxs for d in devs], [np.zeros(len(d.ctrl_idxs)) for d in devs]

    def calc_error(self, target, device):
        return self.real.calc_error(target, device)


def _runner(rate=0.2):
    dyn = ToyDynamics(rate=rate)
    sim = randomize(FakeSim(free_joint_names=FREE, dynamics=dyn), np.random.default_rng(1))
    app = ic.MujocoApp("default_xyz_abg.yaml", None, sim=sim)
    robot = app.get_robot("DualUR5")
    real = ic.OSC.__new__(ic.OSC)                       # calc_error only needs no GPU state
    ctrl = ZeroController(robot, real)
    r = ActionSequenceRunner(app, ctrl, active_arm="left")
    ee = {"ur5right": "ur_EE_ur5right", "ur5left": "ur_EE_ur5left"}
    dyn.goal_provider = lambda: ({ee[n]: t.get_xyz() for n, t in r.targets.items()},
                                 {ee[n]: t.get_quat() for n, t in r.targets.items()})
    return sim, r, ctrl


def test_config_and_object_relative_waypoint():
    cfg = load_action_config()
    assert [e["action"] for e in cfg["insertion_action_sequence"]].count("WP") == 8
    sim, r, ctrl = _runner()
    r.action_objects = cfg["grommet_action_objects"]
    r.initialize_action_objects()
    male = cfg["grommet_action_objects"]["male_object"]
    assert np.allclose(sim.data.get_joint_qpos(male["joint_name"])[:3], male["initial_pos_xyz"])
    r.set_waypoint_targets(dict(action="WP", target_xyz="male_object", target_abg="male_object", offset="hover_offset"))
    assert np.allclose(r.targets["ur5left"].get_xyz(), np.array(male["initial_pos_xyz"]) + male["hover_offset"])
    assert np.allclose(r.targets["ur5right"].get_quat(), DEFAULT_EE_QUAT)           # passive arm holds its pose
    assert np.allclose(r.targets["ur5right"].get_xyz(), r.ur5right.get_state(DeviceState.EE_XYZ))


def test_waypoint_adapts_max_vel_and_grip_counts_ticks():
    sim, r, ctrl = _runner()
    r.action_objects = load_action_config()["nist_action_objects"]
    r.initialize_action_objects()
    seen = []
    r.on_tick = lambda rr, f: seen.append(rr.active_arm.max_vel[0])
    r.run_sequence([dict(action="WP", target_xyz="male_object", target_abg="male_object", offset="grip_offset", max_speed_xyz=0.7),
                    dict(action="GRIP", gripper_force=0.2, gripper_duration=0.005)])
    n_wp = len(seen) - 5
    assert n_wp > 5 and r.errors["ur5left"] <= 0.0018 * 1.3
    assert seen[0] == 0.7                                      # first tick: error = inf -> clipped to max_speed_xyz
    assert min(seen[:n_wp]) == 0.1                             # near the goal: kp * error under min_speed_xyz
    assert all(0.1 <= v <= 0.7 for v in seen)
    assert sim.data.ctrl[GRIPPER_CTRL_IDX["ur5left"]] == 0.2 and ctrl.calls == len(seen) == r.ticks


class _StubBatched:
    """What FleetActionSequenceRunner needs of BatchedOSC, on the CPU: records whose ee_pose is a state the test moves."""

    def __init__(self, lay, B):
        self.layout, self.B = lay, B
        self.ee = np.zeros((B, lay.ndev, 7))
        self.ee[:, :, 3] = 1.0
        self.gains, self.targets = None, None

    def upload_q(self, q, qd): pass
    def frontend(self): pass
    def download_records(self, keys=("ee_pose",)):
        assert tuple(keys) == ("ee_pose",)          # the runner itself needs nothing else across PCIe
        return dict(ee_pose=self.ee.copy())
    def set_gains(self, *a): self.gains = a
    def set_targets(self, t): self.targets = np.array(t)
    def step(self): return np.zeros((self.B, self.layout.n))


def test_fleet_runner_state_machine_per_instance():
    """Each instance advances on ITS error: adaptive velocity limits differ per robot, a finished robot holds its target,
    GRIP counts ticks (insertion_task.py:264-318 per instance)."""
    from irl_control_amd import synth
    from irl_control_amd.action_sequence import FleetActionSequenceRunner, _calc_error_batch
    lay = synth.make_layout("k13")
    _, gains, _ = synth.make_batch("k13", 1, seed=0)
    B = 3
    objs = [{"obj": dict(pos=[0.1 * (b + 1), 0.0, 0.0], quat=[1, 0, 0, 0], grip_yaw=0.0, up=[0, 0, 0.05])} for b in range(B)]
    seq = [dict(action="WP", target_xyz="obj", offset="up", max_error=0.01, kp=2.0, min_speed_xyz=0.05, max_speed_xyz=1.0),
           dict(action="GRIP", gripper_force=0.3, gripper_duration=0.003)]
    osc = _StubBatched(lay, B)
    r = FleetActionSequenceRunner(osc, gains, objs, seq, active_arm="right", passive_hold_orientation=True)
    ia = lay.dev_names.index("ur5right")
    q = np.zeros((B, lay.n))
    r.tick(q, q)
    assert np.allclose(osc.targets[:, ia, :3], [[0.1, 0, 0.05], [0.2, 0, 0.05], [0.3, 0, 0.05]])
    # orientation error only: all three are still off by the DEFAULT_EE rotation, so nobody advances
    r.after_step(osc.ee)
    assert (r.action == 0).all() and np.isfinite(r.err).all()
    osc.ee[0, ia] = osc.targets[0, ia]                 # robot 0 arrives
    r.tick(q, q)
    mv = osc.gains[5]
    assert mv.shape == (B, lay.ndev, 2)
    want = [max(0.05, min(1.0, 2.0 * e)) for e in r.err]
    assert np.allclose(mv[:, ia, 0], want)
    r.after_step(osc.ee)
    assert list(r.action) == [1, 0, 0]
    for _ in range(3):
        r.tick(q, q); r.after_step(osc.ee)
    assert list(r.action) == [2, 0, 0] and list(r.done()) == [True, False, False]
    assert r.gripper_force[0] == 0.3
    e = _calc_error_batch(osc.ee[:, ia], osc.targets[:, ia])
    assert np.allclose(e[0], 0) and abs(e[1, 0] + 0.2) < 1e-12
