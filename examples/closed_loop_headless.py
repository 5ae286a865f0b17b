#!/usr/bin/env python3
"""Closed loop without a simulator package: a fleet of Dual-UR5 arms driven to Cartesian targets by the HIP controller,
with the rigid-body front end as the "physics" (what MuJoCo would supply).

Per tick and robot (all B robots at once):
    GPU :  (qpos, qvel) --front end--> M, J, bias, EE pose --OSC step--> u                  (irlosc_frontend + irlosc_step)
    host:  qacc = M^-1 (u - bias);  semi-implicit Euler with dt = 1 ms                       (M, bias read back from HBM)
Every joint is torque-driven (the real scene actuates 15 of the 25), no contacts, no joint limits: enough to show that
controller + front end close the loop - the end effectors converge onto their targets and stay there.

    python examples/closed_loop_headless.py [--robots 16] [--ticks 1500]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from irl_control_amd import BatchedOSC, synth                       # noqa: E402
from irl_control_amd.rigid_body import RigidBodyModel               # noqa: E402


def run(robots=16, ticks=1500, seed=0, dt=1e-3, damping=0.0, verbose=True):
    rng = np.random.default_rng(seed)
    lay = synth.make_layout("k13")
    _, gains, _ = synth.make_batch("k13", 1, seed=0)
    model = RigidBodyModel.load("dual_ur5")
    osc = BatchedOSC(lay, robots, dtype=np.float64)
    osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
    osc.set_model(model)
    # start and goal configurations: moderate arm angles, goal = start + a perturbation (so the targets are reachable)
    q = np.zeros((robots, 25))
    q[:, 1:7] = rng.uniform(-1.0, 1.0, (robots, 6)) + np.array([0.0, -0.6, 1.2, 0.0, 0.8, 0.0])
    q[:, 13:19] = rng.uniform(-1.0, 1.0, (robots, 6)) + np.array([0.0, -0.6, 1.2, 0.0, 0.8, 0.0])
    q_goal = q.copy()
    q_goal[:, 1:7] += rng.uniform(-0.35, 0.35, (robots, 6))
    q_goal[:, 13:19] += rng.uniform(-0.35, 0.35, (robots, 6))
    q_goal[:, 0] += rng.uniform(-0.3, 0.3, robots)
    qd = np.zeros_like(q)
    osc.upload_q(q_goal, qd)
    osc.frontend()
    tgt = osc.download_records()["ee_pose"].copy()           # EE poses at the goal configuration = the targets
    err0 = None
    hist = []
    for t in range(ticks):
        # the front end plays the simulator here: its records (M, bias, EE poses) are read back for the host-side physics, so
        # it runs as a kernel of its own and the controller steps on the records (osc.step_from_q would take the fused path,
        # which leaves no dense records behind)
        osc.upload_q(q, qd)
        osc.frontend()
        osc.set_targets(tgt)
        u = osc.step()
        rec = osc.download_records()
        pos_err = np.linalg.norm(rec["ee_pose"][:, :2, :3] - tgt[:, :2, :3], axis=2)      # the two arms
        if err0 is None:
            err0 = pos_err.copy()
        hist.append(pos_err.max())
        qacc = np.linalg.solve(rec["M"], (u - rec["bias"] - damping * qd)[:, :, None])[:, :, 0]
        qd = qd + dt * qacc
        q = q + dt * qd
    osc.close()
    final = pos_err
    if verbose:
        print(f"{robots} robots, {ticks} ticks: EE position error {err0.mean():.3f} m (max {err0.max():.3f}) -> "
              f"{final.mean():.4f} m (max {final.max():.4f})")
    return dict(err0=err0, err=final, hist=np.array(hist), q=q, qd=qd)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--robots", type=int, default=16)
    ap.add_argument("--ticks", type=int, default=1500)
    a = ap.parse_args()
    run(a.robots, a.ticks)
