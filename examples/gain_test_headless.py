#!/usr/bin/env python3
"""Headless counterpart of the reference's gain_test tick loop (examples/gain_test.py:98-175 in
ir-lab/irl_control): same objects, same per-tick calls, no viewer and no MuJoCo.

    targets -> OSC.generate(targets) -> sim.data.ctrl[idx] = force -> waypoint switching -> sim.step()

The simulator is injected (`MujocoApp(cfg, scene, sim=...)`): an `MjSim` works as it is; here a `FakeSim` stands in,
with a toy "dynamics" that slides each end effector a little towards its target per tick so that the waypoint logic
has something to do.  OSC.generate runs on the GPU through libirlosc (B = 1); there is no CPU fallback.

    python examples/gain_test_headless.py [--ticks 200] [--demo gain_test|figure8]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import irl_control_amd as irl_control                      # noqa: E402  (drop-in for `import irl_control`)
from irl_control_amd.device import DeviceState             # noqa: E402
from irl_control_amd.fakesim import FakeSim, randomize     # noqa: E402
from irl_control_amd.utils import Target                   # noqa: E402

THRESHOLD_EE = 0.1                                         # examples/gain_test.py:118


def waypoint_path():
    """The back-and-forth path of the reference demo (examples/gain_test.py:80-96)."""
    return np.array([[0.8, 0.6, 0.7], [0.8, -0.6, 0.7]]), np.array([[-0.5, -0.5, 0.5]])


def figure_eight_path():
    """Closed polyline through six corner points per arm, four interpolated points per edge, mirrored for the right arm."""
    corners = np.array([[-0.8, -0.4, 0.5], [-0.9, -0.35, 0.7], [-0.9, -0.2, 0.5],
                        [-0.9, -0.6, 0.2], [-0.7, -0.7, 0.2], [-0.4, -0.8, 0.3]])
    closed = np.vstack([corners, corners[:1]])
    left = np.vstack([np.linspace(closed[i], closed[i + 1], 5) for i in range(len(corners))])
    right = left.copy()
    right[:, :2] *= -1
    return right, left


class SlideTowardsTargets:
    """Toy stand-in for physics: every step moves each end effector 5 % of the way to its current target."""

    def __init__(self):
        self.goal = {}

    def __call__(self, sim, integrate=False):
        if not integrate:
            return
        for body, xyz in self.goal.items():
            b = sim.model.body_name2id(body)
            sim.data.body_xpos[b] += 0.05 * (np.asarray(xyz) - sim.data.body_xpos[b])


def run(ticks=200, demo="gain_test", seed=0, robot_config="default_xyz.yaml", verbose=True):
    dyn = SlideTowardsTargets()
    sim = randomize(FakeSim(dynamics=dyn), np.random.default_rng(seed))
    app = irl_control.MujocoApp(robot_config, None, sim=sim)
    robot = app.get_robot("DualUR5")
    cfgs = [("base", app.get_controller_config("osc0")), ("ur5right", app.get_controller_config("osc2")),
            ("ur5left", app.get_controller_config("osc2"))]
    controller = irl_control.OSC(robot, sim, cfgs, app.get_controller_config("nullspace"))
    right_wps, left_wps = waypoint_path() if demo == "gain_test" else figure_eight_path()
    targets = {"ur5right": Target(), "ur5left": Target(), "base": Target()}      # dict order = row / output order
    ur5right, ur5left = robot.get_device("ur5right"), robot.get_device("ur5left")
    ri = li = 0
    switches = 0
    for tick in range(ticks):
        targets["ur5right"].set_xyz(right_wps[ri])
        targets["ur5left"].set_xyz(left_wps[li])
        for force_idx, force in zip(*controller.generate(targets)):
            sim.data.ctrl[force_idx] = force
        err_r = np.linalg.norm(ur5right.get_state(DeviceState.EE_XYZ) - targets["ur5right"].get_xyz())
        err_l = np.linalg.norm(ur5left.get_state(DeviceState.EE_XYZ) - targets["ur5left"].get_xyz())
        if err_r < THRESHOLD_EE:
            ri = (ri + 1) % len(right_wps)
            switches += 1
        if err_l < THRESHOLD_EE:
            li = (li + 1) % len(left_wps)
            switches += len(left_wps) > 1
        dyn.goal = {"ur_EE_ur5right": right_wps[ri], "ur_EE_ur5left": left_wps[li]}
        sim.step()
    if verbose:
        print(f"{ticks} ticks, {switches} waypoint switches, |ctrl|max = {np.abs(sim.data.ctrl).max():.3g}")
    return dict(switches=switches, ctrl=np.array(sim.data.ctrl), errors=(err_r, err_l))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--demo", default="gain_test", choices=["gain_test", "figure8"])
    a = ap.parse_args()
    run(a.ticks, a.demo)
