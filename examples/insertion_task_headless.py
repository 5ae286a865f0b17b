#!/usr/bin/env python3
"""Headless counterpart of the reference's insertion demo (examples/insertion_task.py in ir-lab/irl_control): the
WP / GRIP action sequence of action_sequence_configs/insertion_task.yaml run by irl_control_amd.ActionSequenceRunner
(error-adaptive max_vel, object-relative waypoints) with OSC.generate on the HIP path every tick.

Simulator injected: FakeSim with the four free joints of insertion_task_scene.xml and ToyDynamics sliding each end
effector towards its current target (no contacts, so the grasp is only actuated, not simulated).

    python examples/insertion_task_headless.py [--arm right|left] [--objects nist_action_objects] [--with-grip]
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

import irl_control_amd as irl_control                      # noqa: E402
from irl_control_amd.action_sequence import ActionSequenceRunner, load_action_config   # noqa: E402
from irl_control_amd.fakesim import FakeSim, ToyDynamics, randomize     # noqa: E402

FREE_JOINTS = ["free_joint_grommet_11mm", "free_joint_dual_peg", "free_joint_female", "free_joint_male"]
EE_BODY = {"ur5right": "ur_EE_ur5right", "ur5left": "ur_EE_ur5left"}


def build(seed=0, active_arm="right", rate=0.08, dtype=np.float64, on_tick=None, tick_seconds=0.001):
    dyn = ToyDynamics(rate=rate)
    sim = randomize(FakeSim(free_joint_names=FREE_JOINTS, dynamics=dyn), np.random.default_rng(seed))
    app = irl_control.MujocoApp("default_xyz_abg.yaml", None, sim=sim)
    robot = app.get_robot("DualUR5")
    cfgs = [("base", app.get_controller_config("osc0")), ("ur5right", app.get_controller_config("osc2")),
            ("ur5left", app.get_controller_config("osc2"))]
    controller = irl_control.OSC(robot, sim, cfgs, app.get_controller_config("nullspace"), dtype=dtype)
    runner = ActionSequenceRunner(app, controller, active_arm=active_arm, on_tick=on_tick, tick_seconds=tick_seconds)
    dyn.goal_provider = lambda: ({EE_BODY[n]: t.get_xyz() for n, t in runner.targets.items()},
                                 {EE_BODY[n]: t.get_quat() for n, t in runner.targets.items()})
    return sim, runner


def run(seed=0, active_arm="right", objects="nist_action_objects", with_grip=False, rate=0.08, verbose=True, dtype=np.float64,
        tick_seconds=None):
    rec = dict(ctrl=[], max_vel=[])

    def on_tick(r, forces):
        rec["ctrl"].append(np.array(r.sim.data.ctrl))
        rec["max_vel"].append(float(r.active_arm.max_vel[0]))
    sim, runner = build(seed, active_arm, rate, dtype, on_tick, tick_seconds if tick_seconds else 0.001)
    cfg = load_action_config("insertion_task.yaml")
    runner.action_objects = cfg[objects]
    runner.initialize_action_objects()
    seq = cfg["insertion_action_sequence"]
    if not with_grip:
        seq = [e for e in seq if e["action"] == "WP"]
    elif tick_seconds is None:       # keep the example short: the GRIP holds last 20 ticks instead of 1000-2000
        seq = [dict(e, gripper_duration=0.02) if e["action"] == "GRIP" else e for e in seq]
    # (with tick_seconds given, a GRIP lasts gripper_duration / tick_seconds ticks: the golden's 0.04 s -> 25 / 50 ticks)
    runner.run_sequence(seq)
    if verbose:
        print(f"{len(seq)} actions, {runner.ticks} ticks, final error {runner.errors[runner.active_arm.name]:.4g}")
    return dict(ctrl=np.array(rec["ctrl"]), max_vel=np.array(rec["max_vel"]), ticks=runner.ticks, n_actions=len(seq))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--arm", default="right", choices=["right", "left"])
    ap.add_argument("--objects", default="nist_action_objects")
    ap.add_argument("--with-grip", action="store_true")
    a = ap.parse_args()
    run(active_arm=a.arm, objects=a.objects, with_grip=a.with_grip)
