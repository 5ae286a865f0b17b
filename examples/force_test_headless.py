#!/usr/bin/env python3
"""Headless counterpart of the reference's force_test demo (examples/force_test.py in ir-lab/irl_control): admittance
controller with the osc1 gains, the left arm stepping along a line of waypoints 1 cm at a time, the left F/T force read
back after every simulator step (examples/headless_loops.py::force_test_loop; the reference appends it to data.csv).

    python examples/force_test_headless.py [--ticks 240] [--csv data.csv]
"""
import argparse
import csv
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

import irl_control_amd as irl_control                      # noqa: E402
from irl_control_amd.device import DeviceState             # noqa: E402
from irl_control_amd.fakesim import FakeSim, ToyDynamics, randomize     # noqa: E402
from irl_control_amd.utils import Target                   # noqa: E402
import headless_loops as loops                             # noqa: E402


def build(seed=0, robot_config="default_xyz_abg.yaml", dtype=np.float64):
    dyn = ToyDynamics()
    sim = randomize(FakeSim(n_free_bodies=1, dynamics=dyn), np.random.default_rng(seed), wrench=True)
    app = irl_control.MujocoApp(robot_config, None, sim=sim)
    robot = app.get_robot("DualUR5")
    cfgs = [("ur5right", app.get_controller_config("osc1")), ("ur5left", app.get_controller_config("osc1"))]
    controller = irl_control.OSC(robot, sim, cfgs, app.get_controller_config("nullspace"), admittance=True, dtype=dtype)
    return sim, dyn, robot, controller


def run(ticks=240, seed=0, verbose=True, csv_path=None, dtype=np.float64):
    sim, dyn, robot, controller = build(seed, dtype=dtype)
    rec = loops.force_test_loop(robot, controller, Target, DeviceState, sim, ticks, dyn)
    if csv_path:
        with open(csv_path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["x", "force_x", "force_y", "force_z"])      # examples/force_test.py:82-86
            w.writeheader()
            for x, ft in enumerate(rec["ft"], 1):
                w.writerow(dict(x=x, force_x=ft[0], force_y=ft[1], force_z=ft[2]))
    if verbose:
        print(f"{ticks} ticks, left waypoint index reached {int(rec['wp'][:, 1].max())}, |force|max = {np.abs(rec['forces']).max():.3g}")
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=240)
    ap.add_argument("--csv", default=None)
    a = ap.parse_args()
    run(a.ticks, csv_path=a.csv)
