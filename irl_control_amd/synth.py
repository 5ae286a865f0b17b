"""Seeded synthetic Dual-UR5 batches in the C-ABI layout: used by the GPU tests
and by bench.py.  Pure NumPy, no reference, no oracle."""
import numpy as np

from .layout import OSCLayout
from .fakesim import quat_mul, random_unit_quat, synth_jac6, synth_mass_matrix

BASE_J, RIGHT_J, LEFT_J = [0], list(range(1, 13)), list(range(13, 25))
XYZ, ABG, YAW = [True] * 3, [True] * 3, [False, False, True]

LAYOUTS = {
    # name: (targets order, dof masks, joint ids, j_idx0 (rows in sub_devices order base,right,left), flags)
    "k13": (["ur5right", "ur5left", "base"], [XYZ + ABG, XYZ + ABG, [False] * 3 + YAW],
            [RIGHT_J, LEFT_J, BASE_J], [1, 7, 0], dict()),
    "k13_branch_b": (["base", "ur5right", "ur5left"], [[False] * 3 + YAW, XYZ + ABG, XYZ + ABG],
                     [BASE_J, RIGHT_J, LEFT_J], [0, 1, 7], dict(branch_b=True)),
    "k7": (["ur5right", "ur5left", "base"], [XYZ + [False] * 3, XYZ + [False] * 3, [False] * 3 + YAW],
           [RIGHT_J, LEFT_J, BASE_J], [1, 4, 0], dict()),
    # two arms, positions only, no base target: robot_configs/default_xyz.yaml:15-16,24-25 driven with the two arm targets
    "k6": (["ur5right", "ur5left"], [XYZ + [False] * 3, XYZ + [False] * 3], [RIGHT_J, LEFT_J], [1, 4], dict()),
    "k12_admit": (["ur5right", "ur5left"], [XYZ + ABG, XYZ + ABG], [RIGHT_J, LEFT_J], [1, 7],
                  dict(admittance=True)),
}
YAML_GAINS = {  # default_xyz_abg.yaml: base osc0, arms osc2
    "base": dict(kp=2000.0, kv=20.0, ko=2000.0, max_vel=[0.0, 20.0]),
    "ur5right": dict(kp=200.0, kv=50.0, ko=200.0, max_vel=[1.0, 5.0]),
    "ur5left": dict(kp=200.0, kv=50.0, ko=200.0, max_vel=[1.0, 5.0]),
}


def make_layout(cfg: str) -> OSCLayout:
    names, dof, jids, jidx0, fl = LAYOUTS[cfg]
    return OSCLayout(n=25, dev_names=list(names), ctrlr_dof=[list(m) for m in dof],
                     joint_ids=[list(j) for j in jids], j_idx0=list(jidx0),
                     admittance=fl.get("admittance", False))


def targets_near(ee_pose: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """Target poses around given end-effector poses [B, ndev, 7], the recipe of SURVEY.md section 8d: position + N(0, 0.2^2) per
    axis (a mix of velocity-saturated and unsaturated errors), orientation turned by U(0, 0.5 rad) about a random axis."""
    ee = np.asarray(ee_pose, dtype=np.float64)
    B, nd = ee.shape[:2]
    ang = rng.uniform(0.0, 0.5, size=(B, nd, 1))
    ax = rng.normal(size=(B, nd, 3))
    ax /= np.linalg.norm(ax, axis=2, keepdims=True)
    dquat = np.concatenate([np.cos(ang / 2), np.sin(ang / 2) * ax], axis=2)
    return np.concatenate([ee[:, :, :3] + rng.normal(0.0, 0.2, size=(B, nd, 3)), quat_mul(ee[:, :, 3:], dquat)], axis=2)


def make_batch(cfg: str, B: int, seed: int = 0, dtype=np.float64, per_instance_gains: bool = False):
    """-> (layout, gains dict, arrays dict) with arrays in the C-ABI record layout."""
    rng = np.random.default_rng(seed)
    lay = make_layout(cfg)
    names, dof, _, _, fl = LAYOUTS[cfg]
    nd = len(names)
    M = synth_mass_matrix(rng, B)
    J = np.concatenate([synth_jac6(rng, B, nm)[:, np.asarray(m, dtype=bool), :] for nm, m in zip(names, dof)], axis=1)
    dq = rng.normal(0.0, 0.5, size=(B, 25))
    bias = rng.normal(0.0, 5.0, size=(B, 25))
    ee = np.concatenate([rng.uniform(-1, 1, size=(B, nd, 3)), random_unit_quat(rng, B * nd).reshape(B, nd, 4)], axis=2)
    ang = rng.uniform(0.0, 0.5, size=(B, nd, 1))
    ax = rng.normal(size=(B, nd, 3))
    ax /= np.linalg.norm(ax, axis=2, keepdims=True)
    dquat = np.concatenate([np.cos(ang / 2), np.sin(ang / 2) * ax], axis=2)
    tgt = np.concatenate([ee[:, :, :3] + rng.normal(0.0, 0.2, size=(B, nd, 3)), quat_mul(ee[:, :, 3:], dquat)], axis=2)
    arrays = dict(M=M, J=J, dq=dq, bias=bias, ee_pose=ee, tgt_pose=tgt)
    if fl.get("admittance"):
        arrays["wrench"] = rng.normal(0.0, 5.0, size=(B, nd, 6))
    if fl.get("branch_b"):
        tv = rng.normal(0.0, 0.3, size=(B, nd, 6))
        tv[np.abs(tv) < 1e-3] = 0.1
        tv[::3] = 0.0                               # every third instance stays on branch A
        arrays["tgt_vel"] = tv
    if per_instance_gains:
        gains = dict(kp=rng.uniform(100, 2000, (B, nd)), kv=rng.uniform(10, 50, (B, nd)),
                     ko=rng.uniform(50, 2000, (B, nd)), k=rng.uniform(0.5, 3, (B, nd, 3)),
                     d=rng.uniform(0.2, 2, (B, nd, 3)),
                     max_vel=np.array([YAML_GAINS[nm]["max_vel"] for nm in names]), null_kv=rng.uniform(1, 20, B))
    else:
        gains = dict(kp=[YAML_GAINS[nm]["kp"] for nm in names], kv=[YAML_GAINS[nm]["kv"] for nm in names],
                     ko=[YAML_GAINS[nm]["ko"] for nm in names], k=[[1.0, 2.0, 3.0]] * nd, d=[[0.5, 1.0, 1.0]] * nd,
                     max_vel=[YAML_GAINS[nm]["max_vel"] for nm in names], null_kv=10.0)
    arrays = {k: np.ascontiguousarray(v, dtype=dtype) for k, v in arrays.items()}
    return lay, gains, arrays
