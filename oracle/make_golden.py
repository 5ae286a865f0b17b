#!/usr/bin/env python3
"""Mint golden vectors for the OSC hot path by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py          # rewrites tests/golden/*.npz

It imports /root/reference/irl_control with the two import shims of oracle/shims/, builds the
reference's own Device / Robot / OSC objects on a FakeSim (irl_control_amd.fakesim) filled with
seeded synthetic states, calls the reference's ``OSC.generate`` and stores, per case,

  inputs  in the C-ABI record layout of include/irlosc.h (M, stacked J, dq, bias, ee_pose,
          tgt_pose, wrench, tgt_vel, gains, layout descriptor as JSON), extracted through the
          reference's own ``Robot.get_all_states()``;
  outputs the reference's (force_idxs, forces) plus the intermediates of its private ``__Mx``.

Every fixture is data (numbers); no reference source text is stored.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shims"), "/root/reference", ROOT]

import numpy as np  # noqa: E402
import yaml  # noqa: E402

import irl_control as ref  # noqa: E402  (the reference)
from irl_control.device import DeviceState as RefDeviceState  # noqa: E402
from irl_control.robot import RobotState as RefRobotState  # noqa: E402
from irl_control.utils import Target as RefTarget  # noqa: E402

from irl_control_amd import fakesim  # noqa: E402

assert ref.__file__.startswith("/root/reference"), ref.__file__
CFG_DIR = os.path.join(ROOT, "irl_control_amd", "robot_configs")
OUT_DIR = os.environ.get("IRLOSC_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")      # (another directory: tests/test_golden_remint.py)


def load_cfg(name):
    with open(os.path.join(CFG_DIR, name)) as f:
        return yaml.safe_load(f)


def ctrl_cfg(cfg, name):
    import copy
    for e in cfg["controller_configs"]:
        if e["name"] == name:
            return copy.deepcopy(e)
    raise KeyError(name)


def build_reference(cfg, sim, dev_gain_names, nullspace, use_g, admittance, gain_override=None):
    devices = [ref.Device(d, sim.model, sim, True) for d in cfg["devices"]]
    robot = ref.Robot(devices, "DualUR5", sim, True)
    dev_cfgs = []
    for dev_name, gname in dev_gain_names:
        g = ctrl_cfg(cfg, gname)
        if gain_override is not None:
            g.update(gain_override.get(dev_name, {}))
        dev_cfgs.append((dev_name, g))
    ns = ctrl_cfg(cfg, "nullspace") if nullspace else None
    osc = ref.OSC(robot, sim, dev_cfgs, ns, use_g=use_g, admittance=admittance)
    return robot, osc


def quat_about(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([[np.cos(angle / 2)], np.sin(angle / 2) * axis])


def make_case(name, seed, B, cfg_file, target_order, dev_gain_names, nullspace=True, use_g=True,
              admittance=False, all_actuated=False, branch_b=False, degenerate=None, gimbal=False,
              random_gains=False, no_max_vel=(), n_free_bodies=0, dof_override=None):
    """dof_override: {device name: (ctrlr_dof_xyz, ctrlr_dof_abg)} replaces the YAML's row masks (device.py:36) -- the layouts a caller
    reaches with another YAML; target_order may name any subset of the robot's devices (osc.py:134-138 stacks J over the targets)."""
    rng = np.random.default_rng(seed)
    cfg = load_cfg(cfg_file)
    for d in cfg["devices"]:
        if d["name"] in no_max_vel:
            d.pop("max_vel")
        if dof_override and d["name"] in dof_override:
            d["ctrlr_dof_xyz"], d["ctrlr_dof_abg"] = [list(map(bool, m)) for m in dof_override[d["name"]]]
    rec = {k: [] for k in ("M", "J", "dq", "bias", "ee_pose", "tgt_pose", "wrench", "tgt_vel",
                           "kp", "kv", "ko", "kk", "dd", "max_vel", "null_kv",
                           "Mx", "M_inv", "Mx_inv", "det", "forces_flat")}
    layout = None
    force_idxs_ref = None
    for b in range(B):
        if all_actuated:
            names, parent, joints = fakesim.dual_ur5_tree()
            model = fakesim.FakeModel(names, parent, joints,
                                      [j for js in joints for j in js], n_free_bodies)
            sim = fakesim.FakeSim(model)
        else:
            sim = fakesim.FakeSim(n_free_bodies=n_free_bodies)
        fakesim.randomize(sim, rng, wrench=admittance)
        gain_override = None
        if random_gains:
            gain_override = {dn: dict(kp=float(rng.uniform(100, 2000)), kv=float(rng.uniform(10, 50)),
                                      ko=float(rng.uniform(50, 2000)),
                                      k=[float(x) for x in rng.uniform(0.5, 3, 3)],
                                      d=[float(x) for x in rng.uniform(0.2, 2, 3)])
                             for dn, _ in dev_gain_names}
        robot, osc = build_reference(cfg, sim, dev_gain_names, nullspace, use_g, admittance,
                                     gain_override)
        # Device.__init__ overwrote qpos[start angles]; states that matter are the derived arrays.
        if degenerate is not None:
            # make two translational rows of the right arm nearly parallel -> tiny eigenvalue of
            # J M^-1 J^T (pinv-truncation regime when degenerate is small, osc.py:51-55)
            bid = sim.model.body_name2id("ur_EE_ur5right")
            jp = sim.data.body_jacp[bid].reshape(3, -1)
            jp[1] = jp[0] * (1.0 + 0.1 * rng.normal()) + degenerate[b % len(degenerate)] * rng.normal(size=jp.shape[1]) * (jp[0] != 0)
        targets = {}
        for dn in target_order:
            dev = robot.get_device(dn)
            ee_xyz = dev.get_state(RefDeviceState.EE_XYZ)
            ee_quat = dev.get_state(RefDeviceState.EE_QUAT)
            t = RefTarget()
            t.set_xyz(ee_xyz + rng.normal(0.0, 0.2, 3))
            ax = rng.normal(size=3)
            if gimbal and dn != "base":
                # error rotation q_ee (x) conj(q_tgt) with sxyz ay = +-pi/2 (cy ~ 0 branch)
                sgn = 1.0 if b % 2 == 0 else -1.0
                off = [0.0, 1e-9, 1e-7, 1e-5][(b // 2) % 4]
                q_err = fakesim.quat_mul(quat_about([0, 0, 1], rng.uniform(-1, 1)),
                                         fakesim.quat_mul(quat_about([0, 1, 0], sgn * (np.pi / 2 - off)),
                                                          quat_about([1, 0, 0], rng.uniform(-1, 1))))
                # q_ee * conj(q_tgt) = q_err  ->  q_tgt = conj(q_err) * q_ee ... (unit quats)
                q_tgt = fakesim.quat_mul(q_err * np.array([1, -1, -1, -1]), ee_quat)
                # un-normalised target quaternion exercises normalized_vector (osc.py:114)
                t.set_quat(q_tgt * rng.uniform(0.5, 2.0))
            else:
                dq_ = quat_about(ax, rng.uniform(0.0, 0.5 if b % 3 else 3.0))
                t.set_quat(fakesim.quat_mul(ee_quat, dq_) * (1.0 if b % 4 else rng.uniform(0.5, 2.0)))
            if branch_b:
                v = rng.normal(0.0, 0.3, 6)
                v[np.abs(v) < 1e-3] = 0.1
                t.set_xyz_vel(v[:3])
                t.set_abg_vel(v[3:])
            targets[dn] = t
        state = robot.get_all_states()
        Js, J_idxs = state[RefRobotState.J]
        J = np.vstack([Js[dn] for dn in target_order])
        M = state[RefRobotState.M]
        Mx, M_inv = osc._OSC__Mx(J, M)
        Mx_inv = np.dot(J, np.dot(M_inv, J.T))
        force_idxs, forces = osc.generate(targets)

        rec["M"].append(M); rec["J"].append(J); rec["dq"].append(state[RefRobotState.DQ])
        rec["bias"].append(np.array(sim.data.qfrc_bias[robot.joint_ids_all]))
        rec["ee_pose"].append([np.concatenate([state[dn][RefDeviceState.EE_XYZ],
                                               state[dn][RefDeviceState.EE_QUAT]]) for dn in target_order])
        rec["tgt_pose"].append([np.concatenate([targets[dn].get_xyz(), targets[dn].get_quat()])
                                for dn in target_order])
        rec["wrench"].append([np.concatenate([state[dn][RefDeviceState.FORCE],
                                              state[dn][RefDeviceState.TORQUE]]) for dn in target_order])
        rec["tgt_vel"].append([np.hstack([targets[dn].get_xyz_vel(), targets[dn].get_abg_vel()])
                               for dn in target_order])
        dc = osc.device_configs
        rec["kp"].append([dc[dn]["kp"] for dn in target_order])
        rec["kv"].append([dc[dn]["kv"] for dn in target_order])
        rec["ko"].append([dc[dn]["ko"] for dn in target_order])
        rec["kk"].append([dc[dn]["k"] for dn in target_order])
        rec["dd"].append([dc[dn]["d"] for dn in target_order])
        rec["max_vel"].append([(robot.get_device(dn).max_vel or [0, 0]) for dn in target_order])
        rec["null_kv"].append(osc.nullspace_config["kv"] if nullspace else 0.0)
        rec["Mx"].append(Mx); rec["M_inv"].append(M_inv); rec["Mx_inv"].append(Mx_inv)
        rec["det"].append(np.linalg.det(Mx_inv))
        rec["forces_flat"].append(np.concatenate(forces))
        if layout is None:
            force_idxs_ref = [np.asarray(x).tolist() for x in force_idxs]
            layout = dict(
                n=int(robot.num_joints_total), nv=int(sim.model.nv), dev_names=list(target_order),
                dev_rows=[int(Js[dn].shape[0]) for dn in target_order],
                ctrlr_dof=[[bool(x) for x in robot.get_device(dn).ctrlr_dof] for dn in target_order],
                joint_ids=[[int(x) for x in robot.get_device(dn).joint_ids_all] for dn in target_order],
                j_idx0=[int(J_idxs[dn][0]) if len(J_idxs[dn]) else 0 for dn in target_order],
                has_max_vel=[robot.get_device(dn).max_vel is not None for dn in target_order],
                actuator_trnids=[[int(x) for x in robot.get_device(dn).actuator_trnids] for dn in target_order],
                ctrl_idxs=force_idxs_ref,
                use_g=bool(use_g), admittance=bool(admittance), nullspace=bool(nullspace),
                cfg_file=cfg_file, dev_gain_names=[list(x) for x in dev_gain_names],
                seed=seed, branch_b=bool(branch_b), all_actuated=bool(all_actuated))
    arrays = {k: np.asarray(v, dtype=np.float64) for k, v in rec.items()}
    arrays["layout_json"] = np.array(json.dumps(layout))
    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    det = arrays["det"]
    cond = np.array([np.linalg.cond(a) for a in arrays["Mx_inv"]])
    print(f"{name:28s} B={B:3d} k={arrays['J'].shape[1]:2d} |det|<1e-4: {int((np.abs(det) < 1e-4).sum()):3d} "
          f"cond(Mx_inv) med {np.median(cond):.2e} max {cond.max():.2e}  -> {os.path.getsize(path) // 1024} KiB")


def make_e2e_case(name, seed, B, cfg_file, target_order, dev_gain_names, nullspace=True, use_g=True,
                  admittance=False, n_free_bodies=0):
    """End-to-end fixture: the RAW simulator arrays (what a backend holds) plus the reference's
    assembled state and its (force_idxs, forces).  Lets tests rebuild a FakeSim, run the build's own
    MujocoApp/Robot/Device/OSC on it and compare every stage with the reference's."""
    rng = np.random.default_rng(seed)
    cfg = load_cfg(cfg_file)
    ee_bodies = ["ur_stand_dummy", "ur_EE_ur5right", "ur_EE_ur5left"]
    keys = ("qM", "qvel", "qfrc_bias", "sensordata", "jacp", "jacr", "xpos", "xquat", "xmat_right",
            "xmat_left", "tgt_xyz", "tgt_quat", "M", "J", "dq", "wrench", "forces_flat")
    rec = {k: [] for k in keys}
    meta = None
    for b in range(B):
        sim = fakesim.FakeSim(n_free_bodies=n_free_bodies)
        fakesim.randomize(sim, rng, wrench=admittance)
        robot, osc = build_reference(cfg, sim, dev_gain_names, nullspace, use_g, admittance)
        targets = {}
        for dn in target_order:
            dev = robot.get_device(dn)
            t = RefTarget()
            t.set_xyz(dev.get_state(RefDeviceState.EE_XYZ) + rng.normal(0.0, 0.2, 3))
            t.set_abg(rng.uniform(-1.0, 1.0, 3))
            targets[dn] = t
        state = robot.get_all_states()
        Js, J_idxs = state[RefRobotState.J]
        force_idxs, forces = osc.generate(targets)
        d = sim.data
        bids = [sim.model.body_name2id(x) for x in ee_bodies]
        rec["qM"].append(np.array(d.qM)); rec["qvel"].append(np.array(d.qvel))
        rec["qfrc_bias"].append(np.array(d.qfrc_bias)); rec["sensordata"].append(np.array(d.sensordata))
        rec["jacp"].append(d.body_jacp[bids]); rec["jacr"].append(d.body_jacr[bids])
        rec["xpos"].append(d.body_xpos[bids]); rec["xquat"].append(d.body_xquat[bids])
        rec["xmat_right"].append(d.site_xmat["ft_frame_ur5right"]); rec["xmat_left"].append(d.site_xmat["ft_frame_ur5left"])
        rec["tgt_xyz"].append([targets[dn].get_xyz() for dn in target_order])
        rec["tgt_quat"].append([targets[dn].get_quat() for dn in target_order])
        rec["M"].append(state[RefRobotState.M]); rec["dq"].append(state[RefRobotState.DQ])
        rec["J"].append(np.vstack([Js[dn] for dn in target_order]))
        rec["wrench"].append([np.concatenate([state[dn][RefDeviceState.FORCE], state[dn][RefDeviceState.TORQUE]])
                              for dn in target_order])
        rec["forces_flat"].append(np.concatenate(forces))
        if meta is None:
            meta = dict(cfg_file=cfg_file, target_order=list(target_order), ee_bodies=ee_bodies,
                        dev_gain_names=[list(x) for x in dev_gain_names], nullspace=nullspace, use_g=use_g,
                        admittance=admittance, n_free_bodies=n_free_bodies, seed=seed,
                        force_idxs=[np.asarray(x).tolist() for x in force_idxs],
                        force_lens=[len(f) for f in forces],
                        joint_ids={dn: [int(x) for x in robot.get_device(dn).joint_ids] for dn in target_order},
                        joint_names={dn: list(robot.get_device(dn).joint_names) for dn in target_order},
                        J_idxs={dn: [int(x) for x in J_idxs[dn]] for dn in J_idxs})
    arrays = {k: np.asarray(v, dtype=np.float64) for k, v in rec.items()}
    arrays["layout_json"] = np.array(json.dumps(meta))
    path = os.path.join(OUT_DIR, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name:28s} B={B:3d} (end-to-end sim state) -> {os.path.getsize(path) // 1024} KiB")


def make_mutation_case(name, seed, B, cfg_file, target_order, dev_gain_names):
    """Attributes of a LIVE Device changed between generate() calls on one OSC object, the way the reference's examples
    do it: `ctrlr_dof_abg` switched off / on after construction (examples/ps_move_example.py:137-150: only calc_error's
    angle block reacts, osc.py:113; the row mask `ctrlr_dof` was frozen in the constructor, device.py:34-36) and
    `max_vel[0]` re-set before a tick (examples/insertion_task.py:294).  End-to-end format (RAW simulator arrays of one
    state per instance) + per phase the targets and the reference's forces."""
    rng = np.random.default_rng(seed)
    cfg = load_cfg(cfg_file)
    ee_bodies = ["ur_stand_dummy", "ur_EE_ur5right", "ur_EE_ur5left"]
    OFF, ON = [False] * 3, [True] * 3
    phases = [dict(), dict(abg={"ur5right": OFF}), dict(abg={"ur5left": OFF}), dict(abg={"ur5right": ON}),
              dict(abg={"ur5left": ON}, max_vel0={"ur5left": 0.35}), dict(max_vel0={"ur5left": 2.5, "ur5right": 0.1})]
    keys = ("qM", "qvel", "qfrc_bias", "sensordata", "jacp", "jacr", "xpos", "xquat", "xmat_right", "xmat_left",
            "tgt_xyz", "tgt_quat", "forces_flat")
    rec = {k: [] for k in keys}
    meta = None
    for b in range(B):
        cfg = load_cfg(cfg_file)        # fresh per robot: Device keeps the YAML's max_vel LIST, which the phases below mutate
        sim = fakesim.FakeSim()
        fakesim.randomize(sim, rng)
        robot, osc = build_reference(cfg, sim, dev_gain_names, True, True, False)
        d = sim.data
        bids = [sim.model.body_name2id(x) for x in ee_bodies]
        rec["qM"].append(np.array(d.qM)); rec["qvel"].append(np.array(d.qvel))
        rec["qfrc_bias"].append(np.array(d.qfrc_bias)); rec["sensordata"].append(np.array(d.sensordata))
        rec["jacp"].append(d.body_jacp[bids]); rec["jacr"].append(d.body_jacr[bids])
        rec["xpos"].append(d.body_xpos[bids]); rec["xquat"].append(d.body_xquat[bids])
        rec["xmat_right"].append(d.site_xmat["ft_frame_ur5right"]); rec["xmat_left"].append(d.site_xmat["ft_frame_ur5left"])
        txyz, tquat, forces_all = [], [], []
        for ph in phases:
            for dn, mask in ph.get("abg", {}).items():
                robot.get_device(dn).ctrlr_dof_abg = list(mask)
            for dn, v in ph.get("max_vel0", {}).items():
                robot.get_device(dn).max_vel[0] = v
            targets = {}
            for dn in target_order:
                dev = robot.get_device(dn)
                t = RefTarget()
                t.set_xyz(dev.get_state(RefDeviceState.EE_XYZ) + rng.normal(0.0, 0.2, 3))
                t.set_abg(rng.uniform(-1.0, 1.0, 3))
                targets[dn] = t
            force_idxs, forces = osc.generate(targets)
            txyz.append([targets[dn].get_xyz() for dn in target_order])
            tquat.append([targets[dn].get_quat() for dn in target_order])
            forces_all.append(np.concatenate(forces))
        rec["tgt_xyz"].append(txyz); rec["tgt_quat"].append(tquat); rec["forces_flat"].append(forces_all)
        if meta is None:
            meta = dict(cfg_file=cfg_file, target_order=list(target_order), ee_bodies=ee_bodies,
                        dev_gain_names=[list(x) for x in dev_gain_names], nullspace=True, use_g=True, admittance=False,
                        n_free_bodies=0, seed=seed, phases=phases,
                        force_idxs=[np.asarray(x).tolist() for x in force_idxs], force_lens=[len(f) for f in forces])
    arrays = {k: np.asarray(v, dtype=np.float64) for k, v in rec.items()}
    arrays["layout_json"] = np.array(json.dumps(meta))
    path = os.path.join(OUT_DIR, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name:28s} B={B:3d} x {len(phases)} phases (live-device mutations) -> {os.path.getsize(path) // 1024} KiB")


def make_loop_case(name, seed, ticks, kind, cfg_file, dev_gain_names, admittance=False, n_free_bodies=0,
                   push_window=(0, 0)):
    """Per-tick golden of a headless caller loop (harness rows): the REFERENCE's Device / Robot /
    OSC driven by examples/headless_loops.py on a FakeSim with fakesim.ToyDynamics; stores the reference's forces of
    every tick (and the waypoint indices of gain_test).  The GPU tests run the same loop on the build's classes with
    the same seed and compare tick by tick."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import headless_loops as loops
    rng = np.random.default_rng(seed)
    cfg = load_cfg(cfg_file)
    dyn = fakesim.ToyDynamics()
    sim = fakesim.randomize(fakesim.FakeSim(n_free_bodies=n_free_bodies, dynamics=dyn), rng, wrench=admittance)
    robot, osc = build_reference(cfg, sim, dev_gain_names, True, True, admittance)
    if kind == "gain_test":
        rec = loops.gain_test_loop(robot, osc, RefTarget, RefDeviceState, sim, ticks, None, dyn)
    elif kind == "force_test":
        rec = loops.force_test_loop(robot, osc, RefTarget, RefDeviceState, sim, ticks, dyn)
    else:
        rec = loops.admit_test_loop(robot, osc, RefTarget, RefDeviceState, sim, ticks, push_window, dyn)
    meta = dict(kind=kind, seed=seed, ticks=ticks, cfg_file=cfg_file, idxs=rec["idxs"], push_window=list(push_window),
                n_free_bodies=n_free_bodies)
    arrays = dict(forces=np.asarray(rec["forces"], dtype=np.float64), layout_json=np.array(json.dumps(meta)))
    if kind == "gain_test":
        arrays["wp"] = np.asarray(rec["wp"], dtype=np.int64)
        arrays["err"] = np.asarray(rec["err"], dtype=np.float64)
    if kind == "force_test":
        arrays["wp"] = np.asarray(rec["wp"], dtype=np.int64)
        arrays["ft"] = np.asarray(rec["ft"], dtype=np.float64)
    path = os.path.join(OUT_DIR, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    extra = f", {int((np.diff(arrays['wp'], axis=0) != 0).sum())} waypoint switches" if "wp" in arrays else ""
    print(f"{name:28s} {ticks} ticks of the {kind} loop{extra} -> {os.path.getsize(path) // 1024} KiB")


INSERTION_FREE_JOINTS = ["free_joint_grommet_11mm", "free_joint_dual_peg", "free_joint_female", "free_joint_male"]


def insertion_wp_sequence():
    """The WP entries of the reference's insertion action list (the GRIP entries run on a wall-clock timer in the
    reference, insertion_task.py:193-205, so their tick counts are not reproducible and they are left out of the golden)."""
    with open("/root/reference/irl_control/action_sequence_configs/insertion_task.yaml") as f:
        cfg = yaml.safe_load(f)
    return cfg, [e for e in cfg["insertion_action_sequence"] if e["action"] == "WP"]


def make_insertion_golden(name, seed, active_arm="right", objects="nist_action_objects", rate=0.08, with_grip=False,
                          tick_seconds=0.04):
    """Per-tick golden of the action-sequence state machine: the REFERENCE's own
    InsertionTask methods (set_waypoint_targets, go_to_waypoint, grip, send_forces, initialize_action_objects,
    run_sequence; examples/insertion_task.py:146-318) on a FakeSim with free joints and ToyDynamics.  The object is
    created without its constructor (which loads MuJoCo and opens a viewer); everything it would have set up is supplied
    here.  with_grip: the whole action list, GRIP entries included.  The reference's `grip` (insertion_task.py:190-205)
    loops `while self.timer_running` around a thread that sleeps for `gripper_duration` of WALL-CLOCK time; here the
    thread is replaced by a tick budget -- start() raises timer_running, every rendered tick consumes one tick of
    gripper_duration / tick_seconds, the last one lowers the flag -- so the reference's own loop body runs a stated,
    reproducible number of ticks (what ActionSequenceRunner.grip does by construction)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_insertion_task", "/root/reference/irl_control/examples/insertion_task.py")
    it_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(it_mod)
    from mujoco_py.mjviewer import MjViewer
    rng = np.random.default_rng(seed)
    cfg = load_cfg("default_xyz_abg.yaml")
    dyn = fakesim.ToyDynamics(rate=rate)
    sim = fakesim.randomize(fakesim.FakeSim(free_joint_names=INSERTION_FREE_JOINTS, dynamics=dyn), rng)
    robot, osc = build_reference(cfg, sim, G_GAIN, True, True, False)
    it = it_mod.InsertionTask.__new__(it_mod.InsertionTask)
    it.sim, it.model, it.robot, it.controller = sim, sim.model, robot, osc
    it.ur5right, it.ur5left = robot.get_device("ur5right"), robot.get_device("ur5left")
    it.set_active_arm(active_arm)
    it.errors = dict()
    it.viewer = MjViewer(sim)
    it.action_map = it.get_action_map()
    it.DEFAULT_PARAMS = dict([(a, it.get_default_action_ctrl_params(a)) for a in it_mod.Action])
    it.targets = {it.active_arm.name: RefTarget(), it.passive_arm.name: RefTarget()}
    it.timer_running = False
    acfg, seq = insertion_wp_sequence()
    budget = dict(left=0)
    if with_grip:
        seq = acfg["insertion_action_sequence"]

        class TickBudgetThread:                                  # stands in for threading.Thread(target=self.sleep_for, args=(t,))
            def __init__(self, target=None, args=()):
                self.seconds = float(args[0])

            def start(self):
                assert it.timer_running is False                 # mujoco_app.py:38
                budget["left"] = max(1, int(round(self.seconds / tick_seconds)))
                it.timer_running = True
        it_mod.threading = type("threading_stub", (), {"Thread": TickBudgetThread})
    it.action_objects = acfg[objects]
    it.initialize_action_objects()
    ee = {"ur5right": "ur_EE_ur5right", "ur5left": "ur_EE_ur5left"}
    dyn.goal_provider = lambda: ({ee[n]: t.get_xyz() for n, t in it.targets.items()},
                                 {ee[n]: t.get_quat() for n, t in it.targets.items()})
    rec = dict(ctrl=[], max_vel=[])

    def on_render():
        rec["ctrl"].append(np.array(sim.data.ctrl))
        rec["max_vel"].append(float(it.active_arm.max_vel[0]))
        if it.timer_running:
            budget["left"] -= 1
            if budget["left"] <= 0:
                it.timer_running = False
    it.viewer.on_render = on_render
    it.run_sequence(seq)
    meta = dict(seed=seed, active_arm=active_arm, objects=objects, rate=rate, free_joints=INSERTION_FREE_JOINTS,
                n_wp=len([e for e in seq if e["action"] == "WP"]), n_actions=len(seq), with_grip=bool(with_grip),
                tick_seconds=tick_seconds, ticks=len(rec["ctrl"]))
    arrays = dict(ctrl=np.asarray(rec["ctrl"]), max_vel=np.asarray(rec["max_vel"]), layout_json=np.array(json.dumps(meta)))
    path = os.path.join(OUT_DIR, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name:28s} {len(seq)} actions ({meta['n_wp']} WP), {len(rec['ctrl'])} ticks -> {os.path.getsize(path) // 1024} KiB")


G_TELEOP = [("base", "osc2"), ("ur5right", "osc2"), ("ur5left", "osc2")]      # both teleoperation demos: osc2 everywhere


class _StopLoop(Exception):
    pass


def _load_reference_example(fname, stubs):
    """Import one of the reference's example scripts with stand-ins for the hardware modules it pulls in."""
    import importlib.util
    import types
    for name, attrs in stubs.items():
        m = types.ModuleType(name)
        for k2, v in attrs.items():
            setattr(m, k2, v)
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("ref_" + fname[:-3], "/root/reference/irl_control/examples/" + fname)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_teleop_golden(name, kind, seed, ticks, rate=0.08, button_every=4):
    """Per-tick goldens of the two teleoperation callers (SURVEY.md section 8b lists both among the consumers of
    OSC.generate): the REFERENCE's own loop bodies -- SpaceMouseDemo.run_demo (examples/space_mouse_example.py:106-165) with
    the reference's own SpaceMouse integrator on a scripted `pyspacemouse`, PSMoveExample.run and its button poll
    (examples/ps_move_example.py:60-180) on scripted MoveState records -- on a FakeSim with ToyDynamics whose goals follow
    the mocap bodies the loops move.  The objects are created without their constructors (MuJoCo, a viewer, the
    controllers' drivers); the viewer's render() is the tick hook.  space_mouse: the wall-clock `sleep_for` thread is a tick
    budget, as for the GRIP actions.  ps_move: `run` never returns -- the hook ends it after `ticks`; its 10 Hz button
    thread runs its own loop body, one pass every `button_every` ticks, handed over through two semaphores where it sleeps."""
    import threading as real_threading
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import teleop_loops as tl
    from mujoco_py.mjviewer import MjViewer
    rng = np.random.default_rng(seed)
    cfg = load_cfg("default_xyz_abg.yaml")
    dyn = fakesim.ToyDynamics(rate=rate)
    mocaps = tl.SPACE_MOUSE_MOCAPS if kind == "space_mouse" else tl.PS_MOVE_MOCAPS
    sim = fakesim.randomize(fakesim.FakeSim(dynamics=dyn, mocap_names=mocaps), rng)
    robot, osc = build_reference(cfg, sim, G_TELEOP, True, True, False)
    ee = {"ur5right": "ur_EE_ur5right", "ur5left": "ur_EE_ur5left"}
    hands = dict(zip(("ur5right", "ur5left"), mocaps[-2:]))
    rec = dict(ctrl=[])
    if kind == "space_mouse":
        dyn.goal_provider = lambda: ({ee[n]: sim.data.get_body_xpos(h).copy() for n, h in hands.items()},
                                     {ee[n]: sim.data.get_body_xquat(h).copy() for n, h in hands.items()})
        mod = _load_reference_example("space_mouse_example.py",
                                      {"pyspacemouse": dict(open=lambda: True, read=tl.space_mouse_stream(seed, ticks))})
        demo = mod.SpaceMouseDemo.__new__(mod.SpaceMouseDemo)
        demo.sim, demo.model, demo.robot, demo.controller = sim, sim.model, robot, osc
        demo.viewer = MjViewer(sim)
        demo.timer_running = False
        budget = dict(left=0)

        class TickBudgetThread:
            def __init__(self, target=None, args=()):
                pass

            def start(self):
                budget["left"] = ticks
                demo.timer_running = True

            def join(self):
                pass
        mod.threading = type("threading_stub", (), {"Thread": TickBudgetThread})

        def on_render():
            rec["ctrl"].append(np.array(sim.data.ctrl))
            budget["left"] -= 1
            if budget["left"] <= 0:
                demo.timer_running = False
        demo.viewer.on_render = on_render
        demo.run_demo(demo_duration=ticks)
    else:
        dyn.goal_provider = lambda: ({ee[n]: sim.data.get_body_xpos(h).copy() for n, h in hands.items()}, {})
        mod = _load_reference_example("ps_move_example.py", {"psmove": dict(count_connected=lambda: 0)})
        RefMoveName = mod.MoveName
        from irl_control.input_devices.ps_move import MoveState as RefMoveState
        script = tl.ps_move_script(seed, ticks + 1)
        by_ref = {RefMoveName.RIGHT: tl.MoveName.RIGHT, RefMoveName.LEFT: tl.MoveName.LEFT}
        demo = mod.PSMoveExample.__new__(mod.PSMoveExample)
        demo.sim, demo.model, demo.robot, demo.controller = sim, sim.model, robot, osc
        demo.viewer = MjViewer(sim)
        demo.move_states = {n: RefMoveState() for n in RefMoveName}
        demo.grip_pos = dict([(n, 0.0) for n in RefMoveName])

        def apply_row(t):
            for rn, n in by_ref.items():
                for k2, v in script[t][n].items():
                    demo.move_states[rn].set(k2, v)
        apply_row(0)
        go, done, state = real_threading.Semaphore(0), real_threading.Semaphore(0), dict(stop=False, tick=0)

        class ButtonThread:                                  # stands in for threading.Thread(target=self.update_move_button_states)
            def __init__(self, target=None, args=()):
                def body():
                    go.acquire()                             # no pass before the loop asks for one
                    if not state["stop"]:
                        target(*args)
                self._t = real_threading.Thread(target=body, daemon=True)

            def start(self):
                self._t.start()

        def sleep_stub(_seconds):                            # the poll's time.sleep: one pass is over, wait to be asked again
            done.release()
            go.acquire()
            if state["stop"]:
                raise SystemExit
        mod.threading = type("threading_stub", (), {"Thread": ButtonThread})
        mod.time = type("time_stub", (), {"sleep": staticmethod(sleep_stub)})

        def on_render():
            rec["ctrl"].append(np.array(sim.data.ctrl))
            t = state["tick"]
            if (t + 1) % button_every == 0:
                go.release()
                done.acquire()
            state["tick"] = t + 1
            if t + 1 >= ticks:
                raise _StopLoop
            apply_row(t + 1)
        demo.viewer.on_render = on_render
        try:
            demo.run()
        except _StopLoop:
            pass
        state["stop"] = True
        go.release()
    # cross-check of the headless counterpart's LOGIC: examples/teleop_loops.py driven with the reference's classes on the same
    # set-up must reproduce the reference's own loop (the GPU tests then swap in the HIP path) -- to rounding, not bit for bit: the
    # reference's loop body computes its rotations through oracle/shims/transforms3d (SciPy), the headless counterpart through the
    # product's own restatement (irl_control_amd/transforms.py); a LOGIC slip (a trigger, a mask, a waypoint) moves ctrl by O(1)
    dyn2 = fakesim.ToyDynamics(rate=rate)
    sim2 = fakesim.randomize(fakesim.FakeSim(dynamics=dyn2, mocap_names=mocaps), np.random.default_rng(seed))
    robot2, osc2 = build_reference(load_cfg("default_xyz_abg.yaml"), sim2, G_TELEOP, True, True, False)
    if kind == "space_mouse":
        dyn2.goal_provider = lambda: ({ee[n]: sim2.data.get_body_xpos(h).copy() for n, h in hands.items()},
                                      {ee[n]: sim2.data.get_body_xquat(h).copy() for n, h in hands.items()})
        from irl_control_amd.input_devices import SpaceMouse
        mine = tl.space_mouse_loop(robot2, osc2, RefTarget, sim2, SpaceMouse([0.0, 0.5, 0.5, 0.0, 0.0, 0.0],
                                                                             reader=tl.space_mouse_stream(seed, ticks)), ticks)
    else:
        dyn2.goal_provider = lambda: ({ee[n]: sim2.data.get_body_xpos(h).copy() for n, h in hands.items()}, {})
        from irl_control_amd.input_devices import MoveState
        ms = {n: MoveState() for n in tl.MoveName}
        script2 = tl.ps_move_script(seed, ticks + 1)
        tl.apply_script_row(ms, script2[0])
        mine = tl.ps_move_loop(robot2, osc2, RefTarget, RefDeviceState, sim2, ms, ticks,
                               advance=lambda t: tl.apply_script_row(ms, script2[t + 1]), button_every=button_every)
    _d = np.abs(mine["ctrl"] - np.asarray(rec["ctrl"]))
    print("   self-check:", _d.max(), "first tick differing:", int(np.argmax(_d.max(axis=1) > 0)), "cols", np.nonzero(_d.max(axis=0) > 0)[0])
    assert _d.max() <= 1e-9 * max(1.0, float(np.abs(np.asarray(rec["ctrl"])).max())), "examples/teleop_loops.py does not reproduce the reference's loop"
    meta = dict(kind=kind, seed=seed, ticks=ticks, rate=rate, button_every=button_every, mocaps=list(mocaps))
    arrays = dict(ctrl=np.asarray(rec["ctrl"]), layout_json=np.array(json.dumps(meta)))
    path = os.path.join(OUT_DIR, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name:28s} {len(rec['ctrl'])} ticks of the reference's {kind} loop -> {os.path.getsize(path) // 1024} KiB")


RLB = ("ur5right", "ur5left", "base")
BRL = ("base", "ur5right", "ur5left")
G_GAIN = [("base", "osc0"), ("ur5right", "osc2"), ("ur5left", "osc2")]
G_ADMIT = [("ur5right", "osc2"), ("ur5left", "osc2")]

if __name__ == "__main__":
    S = 20241008
    # `--only a,b` re-mints just those fixtures (every fixture is a pure function of its arguments, so the full run gives
    # the same arrays; this only spares rewriting the zip containers of the others)
    if "--only" in sys.argv:
        _only = set(sys.argv[sys.argv.index("--only") + 1].split(","))

        def _filtered(fn):
            return lambda name, *a, **k: fn(name, *a, **k) if name in _only else None
        make_case, make_e2e_case, make_loop_case = _filtered(make_case), _filtered(make_e2e_case), _filtered(make_loop_case)
        make_insertion_golden, make_mutation_case = _filtered(make_insertion_golden), _filtered(make_mutation_case)
        make_teleop_golden = _filtered(make_teleop_golden)
    # gain_test layout (examples/gain_test.py:27-36,124-128): arms xyz only, k = 7
    make_case("k7_gain_test", S + 1, 32, "default_xyz.yaml", RLB, G_GAIN, all_actuated=True)
    make_case("k7_real_actuators", S + 2, 8, "default_xyz.yaml", RLB, G_GAIN)
    # xyz+abg arms, k = 13 (the 4 096 / 65 536 / 262 144-instance workload)
    make_case("k13_xyz_abg", S + 3, 32, "default_xyz_abg.yaml", RLB, G_GAIN, all_actuated=True)
    make_case("k13_base_first", S + 4, 16, "default_xyz_abg.yaml", BRL, G_GAIN, all_actuated=True)
    make_case("k13_iros2022", S + 5, 16, "iros2022.yaml", RLB, G_GAIN, all_actuated=True)
    # admit_test layout (examples/admit_test.py:19-25,55-58): two arms, admittance, nv = 37
    make_case("k12_admittance", S + 6, 32, "default_xyz_abg.yaml", ("ur5right", "ur5left"), G_ADMIT,
              admittance=True, all_actuated=True, n_free_bodies=2)
    make_case("k13_no_g_no_null", S + 7, 16, "default_xyz_abg.yaml", RLB, G_GAIN, nullspace=False,
              use_g=False, all_actuated=True)
    make_case("k13_branch_b", S + 8, 16, "default_xyz_abg.yaml", BRL, G_GAIN, branch_b=True,
              all_actuated=True)
    make_case("k13_pinv_regime", S + 9, 32, "default_xyz_abg.yaml", RLB, G_GAIN, all_actuated=True,
              degenerate=[1e-2, 1e-3, 1e-4, 3e-6, 1e-6, 1e-7, 0.0, 3e-3])
    make_case("k13_gimbal", S + 10, 16, "default_xyz_abg.yaml", RLB, G_GAIN, all_actuated=True, gimbal=True)
    make_case("k13_random_gains", S + 11, 16, "default_xyz_abg.yaml", RLB, G_GAIN, all_actuated=True,
              random_gains=True)
    make_e2e_case("e2e_gain_test", S + 21, 6, "default_xyz.yaml", RLB, G_GAIN)
    make_e2e_case("e2e_admit_test", S + 22, 6, "default_xyz_abg.yaml", ("ur5right", "ur5left"), G_ADMIT,
                  admittance=True, n_free_bodies=2)
    # per-tick goldens of the two caller loops (examples/gain_test.py:98-175, examples/admit_test.py:43-80)
    make_loop_case("loop_gain_test", 0, 160, "gain_test", "default_xyz.yaml", G_GAIN)
    make_loop_case("loop_admit_test", 0, 120, "admit_test", "default_xyz_abg.yaml", G_ADMIT, admittance=True,
                   n_free_bodies=2, push_window=(40, 80))
    make_loop_case("loop_force_test", 0, 240, "force_test", "default_xyz_abg.yaml",
                   [("ur5right", "osc1"), ("ur5left", "osc1")], admittance=True, n_free_bodies=1)
    make_insertion_golden("loop_insertion_wp", 0)
    make_case("k13_no_max_vel", S + 12, 8, "default_xyz_abg.yaml", RLB, G_GAIN, all_actuated=True,
              no_max_vel=("ur5left",))
    # reference behaviours that only show when a live Device is changed between ticks (ps_move_example.py:137-150,
    # insertion_task.py:294), and the GRIP actions of the insertion sequence on a stated tick budget
    make_mutation_case("e2e_live_mutations", S + 23, 6, "default_xyz_abg.yaml", RLB, G_GAIN)
    make_insertion_golden("loop_insertion_full", 0, with_grip=True, tick_seconds=0.04)
    # the two teleoperation callers of OSC.generate (examples/space_mouse_example.py, examples/ps_move_example.py): the
    # reference's own loop bodies on scripted input streams
    make_teleop_golden("loop_space_mouse", "space_mouse", 5, 200)
    make_teleop_golden("loop_ps_move", "ps_move", 6, 200)
    # round 5: layouts beyond the four of the shipped examples -- target SUBSETS and other row masks -- which the HIP path runs on the
    # KMAX-padded row16 kernels: the reference's own answers for them (its OSC.generate stacks J over whatever targets it is handed)
    T, F = True, False
    make_case("k6_single_arm", S + 31, 16, "default_xyz_abg.yaml", ("ur5right",), [("ur5right", "osc2")], all_actuated=True)
    make_case("k3_single_arm_xyz", S + 32, 16, "default_xyz.yaml", ("ur5left",), [("ur5left", "osc2")], all_actuated=True)
    make_case("k7_base_and_arm", S + 33, 16, "default_xyz_abg.yaml", ("base", "ur5left"), [("base", "osc0"), ("ur5left", "osc2")],
              all_actuated=True)
    make_case("k10_mixed_masks", S + 34, 16, "default_xyz_abg.yaml", RLB, G_GAIN, all_actuated=True,
              dof_override={"ur5left": ([T, T, T], [F, F, F])})
    make_case("k9_admittance_masks", S + 35, 16, "default_xyz_abg.yaml", ("ur5right", "ur5left"), G_ADMIT, admittance=True,
              all_actuated=True, n_free_bodies=2, dof_override={"ur5left": ([T, F, T], [F, T, F])})
    make_e2e_case("e2e_single_arm", S + 37, 6, "default_xyz_abg.yaml", ("ur5right",), [("ur5right", "osc2")], n_free_bodies=1)
    # a device asked for rows no joint can move (the stand only yaws): exact zero rows of J, det = 0, the reference's pinv drops them
    make_case("k15_base_three_rows", S + 36, 16, "default_xyz_abg.yaml", RLB, G_GAIN, all_actuated=True,
              dof_override={"base": ([F, F, F], [T, T, T])})
