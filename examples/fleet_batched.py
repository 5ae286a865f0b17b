#!/usr/bin/env python3
"""A fleet of Dual-UR5 robots controlled per tick by ONE batched OSC call.

Per tick: gather the RAW simulator arrays of all robots (no per-robot Python picking), let the GPU assemble the
controller inputs (`upload_raw` = irlosc_upload_raw: what Robot.get_all_states()/Device.get_state() do, reference
robot.py:44-72 / device.py:115-170), run the operational-space controller for every robot (`step`), and write each
robot's actuator forces back (`u_all[actuator_trnids]` -> `ctrl[ctrl_idxs]`, reference osc.py:203-210).

    python examples/fleet_batched.py [--robots 256] [--ticks 5]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import irl_control_amd as ic                                  # noqa: E402
from irl_control_amd import BatchedOSC, OSCLayout, raw          # noqa: E402
from irl_control_amd.fakesim import FakeSim, randomize          # noqa: E402


def run(n_robots=256, ticks=5, seed=0, dtype=np.float32, verbose=True):
    rng = np.random.default_rng(seed)
    names = ["ur5right", "ur5left", "base"]                     # targets order = row / output order
    sims = [randomize(FakeSim(), rng, wrench=True) for _ in range(n_robots)]
    apps = [ic.MujocoApp("default_xyz_abg.yaml", None, sim=s) for s in sims]
    robots = [a.get_robot("DualUR5") for a in apps]
    devs = [robots[0].get_device(nm) for nm in names]
    lay = OSCLayout.from_devices(devs, robots[0], use_g=True, admittance=False, nullspace=True)
    cfg = {nm: apps[0].get_controller_config("osc0" if nm == "base" else "osc2") for nm in names}
    osc = BatchedOSC(lay, n_robots, dtype=dtype)
    osc.set_gains(kp=[cfg[nm]["kp"] for nm in names], kv=[cfg[nm]["kv"] for nm in names], ko=[cfg[nm]["ko"] for nm in names],
                  k=[cfg[nm]["k"] for nm in names], d=[cfg[nm]["d"] for nm in names],
                  max_vel=[d.max_vel or [0.0, 0.0] for d in devs], null_kv=apps[0].get_controller_config("nullspace")["kv"])
    desc = raw.raw_desc(robots[0], names, np.size(sims[0].data.sensordata))
    t_asm = t_ctl = 0.0
    for tick in range(ticks):
        arrs = raw.collect_raw(sims, robots, names)
        ee = np.concatenate([arrs["ee_xpos"], arrs["ee_xquat"]], axis=2)
        tgt = ee.copy()
        tgt[:, :, :3] += 0.1 * np.sin(0.1 * tick + np.arange(3))     # everybody chases a slowly moving offset
        t0 = time.perf_counter()
        osc.upload_raw(desc, **arrs)
        osc.set_targets(tgt)
        t1 = time.perf_counter()
        u = osc.step()
        t2 = time.perf_counter()
        t_asm += t1 - t0
        t_ctl += t2 - t1
        for sim, rob, ui in zip(sims, robots, u):
            for nm in names:
                dv = rob.get_device(nm)
                sim.data.ctrl[dv.ctrl_idxs] = ui[dv.actuator_trnids]
            sim.step()
    osc.close()
    if verbose:
        print(f"{n_robots} robots x {ticks} ticks: upload_raw+targets {1e3 * t_asm / ticks:.2f} ms/tick, "
              f"controller {1e3 * t_ctl / ticks:.2f} ms/tick, |ctrl|max {max(np.abs(s.data.ctrl).max() for s in sims):.3g}")
    return np.stack([s.data.ctrl for s in sims])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--robots", type=int, default=256)
    ap.add_argument("--ticks", type=int, default=5)
    a = ap.parse_args()
    run(a.robots, a.ticks)
