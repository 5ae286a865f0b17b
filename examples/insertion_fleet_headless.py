#!/usr/bin/env python3
"""The insertion action sequence for a FLEET ("batched over randomised object poses"): B
Dual-UR5 robots, each with its own randomly placed male / female objects (the ranges of
examples/insertion_task.py:343-372 in ir-lab/irl_control), run the WP / GRIP list of insertion_task.yaml in lockstep:
one batched GPU tick per control tick (rigid-body front end + OSC step, per-instance error-adaptive velocity limits),
host-side integration of M qacc = u - bias with M and bias read back from HBM (no contacts: the objects are waypoints).

    python examples/insertion_fleet_headless.py [--robots 32] [--max-ticks 6000]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from irl_control_amd import BatchedOSC, synth                                         # noqa: E402
from irl_control_amd.action_sequence import FleetActionSequenceRunner, load_action_config   # noqa: E402
from irl_control_amd.rigid_body import RigidBodyModel                                 # noqa: E402
from irl_control_amd.transforms import euler2quat                                     # noqa: E402


def random_objects(rng, lay, model, q0, cfg_objects):
    """Per robot: the action objects of the YAML, placed where the active arm can certainly take them: the pose of an
    object is chosen such that its grip waypoint (position + the orientation rule of insertion_task.py:249-262) is the
    end-effector pose of a randomly perturbed joint configuration (forward kinematics by the GPU front end)."""
    from irl_control_amd.action_sequence import DEFAULT_EE_ROT
    from irl_control_amd.transforms import euler2mat, mat2euler, quat2mat
    B = len(q0)
    ia = lay.dev_names.index("ur5right")
    probe = BatchedOSC(lay, B, dtype=np.float64)
    probe.set_model(model)
    poses = {}
    for name, spread in (("male_object", 0.35), ("female_object", 0.35)):
        qg = q0.copy()
        qg[:, 1:7] += rng.uniform(-spread, spread, (B, 6))
        probe.upload_q(qg, np.zeros_like(qg)); probe.frontend()
        poses[name] = probe.download_records(keys=("ee_pose",))["ee_pose"][:, ia].copy()
    probe.close()
    out = []
    for b in range(B):
        objs = {}
        for name in ("male_object", "female_object"):
            o = dict(cfg_objects[name])
            E = euler2mat(*(DEFAULT_EE_ROT + [0, 0, np.deg2rad(o["grip_yaw"])]))
            R_obj = quat2mat(poses[name][b, 3:]) @ E.T                       # R_obj E = R_ee(q_g)
            off = o["grip_offset"] if name == "male_object" else o["insert_offset"]
            o["pos"] = poses[name][b, :3] - np.asarray(off)
            o["quat"] = euler2quat(*mat2euler(R_obj))
            objs[name] = o
        out.append(objs)
    return out


SHORT_SEQUENCE = [      # three waypoints whose poses are exactly reachable (each is the FK pose of a nearby configuration)
    dict(action="WP", target_xyz="male_object", target_abg="male_object", offset="grip_offset", max_error=0.01),
    dict(action="GRIP", gripper_force=0.2, gripper_duration=0.02),
    dict(action="WP", target_xyz="female_object", target_abg="female_object", offset="insert_offset", max_error=0.01, max_speed_xyz=1.0),
    dict(action="WP", target_xyz="start_pos", target_abg="female_object", max_speed_xyz=2.0, max_error=0.05),
]


def run(robots=32, max_ticks=12000, seed=0, dt=1e-3, wp_only=True, verbose=True, only=None, sequence=None):
    rng = np.random.default_rng(seed)
    lay = synth.make_layout("k13")
    _, gains, _ = synth.make_batch("k13", 1, seed=0)
    model = RigidBodyModel.load("dual_ur5")
    cfg = load_action_config("insertion_task.yaml")
    seq = sequence if sequence is not None else \
        [dict(e, max_error=max(e.get("max_error", 0.0018), 0.012)) if e["action"] == "WP" else dict(e, gripper_duration=0.05)
         for e in cfg["insertion_action_sequence"] if not (wp_only and e["action"] == "GRIP")]
    B = robots
    q = np.zeros((B, 25))
    q[:, 1:7] = np.array([0.3, -0.3, 1.5, 0.3, 1.1, 0.3]) + rng.uniform(-0.15, 0.15, (B, 6))     # well-conditioned region: cond(J M^-1 J^T) ~ 1e3
    q[:, 13:19] = np.array([-0.2, -0.8, 1.0, -0.2, 0.6, -0.2]) + rng.uniform(-0.15, 0.15, (B, 6))
    qd = np.zeros_like(q)
    objects = random_objects(rng, lay, model, q, cfg["nist_action_objects"])
    sel = np.arange(B) if only is None else np.asarray(only)
    q, qd, objects = q[sel], qd[sel], [objects[i] for i in sel]
    osc = BatchedOSC(lay, len(sel), dtype=np.float64)
    osc.set_model(model)
    runner = FleetActionSequenceRunner(osc, gains, objects, seq, active_arm="right", passive_hold_orientation=True,
                                       integrator_records=("M", "bias"))      # what the toy physics below reads
    traj = []
    while not runner.done().all() and runner.ticks < max_ticks:
        u, rec = runner.tick(q, qd)
        qacc = np.linalg.solve(rec["M"], (u - rec["bias"])[:, :, None])[:, :, 0]
        qd = qd + dt * qacc
        q = q + dt * qd
        osc.upload_q(q, qd); osc.frontend()
        runner.after_step(osc.download_records(keys=("ee_pose",))["ee_pose"].astype(np.float64))
        traj.append(u.copy())
    osc.close()
    if verbose:
        print(f"{len(sel)} robots, {len(seq)} actions each: {int(runner.done().sum())} finished in {runner.ticks} ticks "
              f"(actions reached: min {runner.action.min()}, max {runner.action.max()})")
    return dict(done=runner.done(), action=runner.action.copy(), ticks=runner.ticks, u=np.array(traj), q=q)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--robots", type=int, default=32)
    ap.add_argument("--max-ticks", type=int, default=12000)
    ap.add_argument("--full", action="store_true", help="the full insertion list (hover waypoints included: without contacts and joint "
                    "limits several of them end in ill-conditioned poses where the reference's pinv cut stalls the arm) instead of the short one")
    a = ap.parse_args()
    run(a.robots, a.max_ticks, sequence=None if a.full else SHORT_SEQUENCE)
