#!/usr/bin/env python3
"""Headless counterpart of the reference's gain_test demo (examples/gain_test.py in ir-lab/irl_control): same
objects, same per-tick calls (examples/headless_loops.py::gain_test_loop), no viewer and no MuJoCo.

    targets -> OSC.generate(targets) -> sim.data.ctrl[idx] = force -> waypoint switching -> sim.step()

The simulator is injected (`MujocoApp(cfg, scene, sim=...)`): an `MjSim` works as it is; here a `FakeSim` stands in,
with `fakesim.ToyDynamics` sliding each end effector towards its target so that the waypoint logic has something to
do.  OSC.generate runs on the GPU through libirlosc (one irlosc_tick per call); there is no CPU fallback.

    python examples/gain_test_headless.py [--ticks 200] [--demo gain_test|figure8]
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

import irl_control_amd as irl_control                      # noqa: E402  (drop-in for `import irl_control`)
from irl_control_amd.device import DeviceState             # noqa: E402
from irl_control_amd.fakesim import FakeSim, ToyDynamics, randomize     # noqa: E402
from irl_control_amd.utils import Target                   # noqa: E402
import headless_loops as loops                             # noqa: E402


def build(seed=0, robot_config="default_xyz.yaml", dtype=np.float64):
    dyn = ToyDynamics()
    sim = randomize(FakeSim(dynamics=dyn), np.random.default_rng(seed))
    app = irl_control.MujocoApp(robot_config, None, sim=sim)
    robot = app.get_robot("DualUR5")
    cfgs = [("base", app.get_controller_config("osc0")), ("ur5right", app.get_controller_config("osc2")),
            ("ur5left", app.get_controller_config("osc2"))]
    controller = irl_control.OSC(robot, sim, cfgs, app.get_controller_config("nullspace"), dtype=dtype)
    return sim, dyn, robot, controller


def run(ticks=200, demo="gain_test", seed=0, robot_config="default_xyz.yaml", verbose=True, dtype=np.float64):
    sim, dyn, robot, controller = build(seed, robot_config, dtype)
    wps = loops.gain_test_waypoints() if demo == "gain_test" else loops.figure_eight_waypoints()
    rec = loops.gain_test_loop(robot, controller, Target, DeviceState, sim, ticks, wps, dyn)
    switches = int((np.diff(rec["wp"], axis=0) != 0).sum())
    if verbose:
        print(f"{ticks} ticks, {switches} waypoint switches, |ctrl|max = {np.abs(sim.data.ctrl).max():.3g}")
    rec.update(switches=switches, ctrl=np.array(sim.data.ctrl), errors=tuple(rec["err"][-1]))
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--demo", default="gain_test", choices=["gain_test", "figure8"])
    a = ap.parse_args()
    run(a.ticks, a.demo)
