#!/usr/bin/env python3
"""Headless counterparts of the reference's two teleoperation demos (examples/space_mouse_example.py and
examples/ps_move_example.py in ir-lab/irl_control): same objects, same per-tick calls (examples/teleop_loops.py), the input
device replaced by a scripted stream, no viewer and no MuJoCo.

    device pose -> targets -> OSC.generate(targets) -> sim.data.ctrl[idx] = force -> mocap bodies -> sim.step()

OSC.generate runs on the GPU through libirlosc (one irlosc_tick per call); there is no CPU fallback.  The PS Move loop
switches `ctrlr_dof_abg` of a live Device with the trigger, so the controller re-keys its layout between ticks.

    python examples/teleop_headless.py [--demo space_mouse|ps_move] [--ticks 200]
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
from irl_control_amd.input_devices import MoveName, MoveState, SpaceMouse   # noqa: E402
from irl_control_amd.utils import Target                   # noqa: E402
import teleop_loops as tl                                  # noqa: E402

EE = {"ur5right": "ur_EE_ur5right", "ur5left": "ur_EE_ur5left"}


def build(demo, seed, rate=0.08, dtype=np.float64):
    mocaps = tl.SPACE_MOUSE_MOCAPS if demo == "space_mouse" else tl.PS_MOVE_MOCAPS
    dyn = ToyDynamics(rate=rate)
    sim = randomize(FakeSim(dynamics=dyn, mocap_names=mocaps), np.random.default_rng(seed))
    app = irl_control.MujocoApp("default_xyz_abg.yaml", None, sim=sim)
    robot = app.get_robot("DualUR5")
    cfgs = [(name, app.get_controller_config("osc2")) for name in ("base", "ur5right", "ur5left")]     # both demos: osc2 everywhere
    controller = irl_control.OSC(robot, sim, cfgs, app.get_controller_config("nullspace"), dtype=dtype)
    hands = dict(zip(("ur5right", "ur5left"), mocaps[-2:]))
    if demo == "space_mouse":                              # the toy physics pulls each end effector towards its mocap hand
        dyn.goal_provider = lambda: ({EE[n]: sim.data.get_body_xpos(h).copy() for n, h in hands.items()},
                                     {EE[n]: sim.data.get_body_xquat(h).copy() for n, h in hands.items()})
    else:
        dyn.goal_provider = lambda: ({EE[n]: sim.data.get_body_xpos(h).copy() for n, h in hands.items()}, {})
    return sim, robot, controller


def run(demo="space_mouse", ticks=200, seed=5, rate=0.08, button_every=4, verbose=True, dtype=np.float64):
    sim, robot, controller = build(demo, seed, rate, dtype)
    if demo == "space_mouse":
        sm = SpaceMouse([0.0, 0.5, 0.5, 0.0, 0.0, 0.0], reader=tl.space_mouse_stream(seed, ticks))     # origin: space_mouse_example.py:117
        rec = tl.space_mouse_loop(robot, controller, Target, sim, sm, ticks)
    else:
        states = {n: MoveState() for n in MoveName}
        script = tl.ps_move_script(seed, ticks + 1)
        tl.apply_script_row(states, script[0])
        rec = tl.ps_move_loop(robot, controller, Target, DeviceState, sim, states, ticks,
                              advance=lambda t: tl.apply_script_row(states, script[t + 1]), button_every=button_every)
    if verbose:
        extra = "" if demo == "space_mouse" else f", trigger changes {int((np.diff(rec['engaged'].astype(int), axis=0) != 0).sum())}"
        print(f"{demo}: {ticks} ticks, |ctrl|max = {np.abs(rec['ctrl']).max():.3g}{extra}")
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--demo", default="space_mouse", choices=["space_mouse", "ps_move"])
    ap.add_argument("--ticks", type=int, default=200)
    a = ap.parse_args()
    run(a.demo, a.ticks, seed=5 if a.demo == "space_mouse" else 6)
