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
ROBOT_ORDER = ["base", "ur5right", "ur5left"]              # robot.sub_devices order: J_idxs count rows in it (robot.py:50-55)
DEV_JOINTS = {"base": BASE_J, "ur5right": RIGHT_J, "ur5left": LEFT_J}
YAML_MASKS = {"base": [False] * 3 + YAW, "ur5right": XYZ + ABG, "ur5left": XYZ + ABG}      # default_xyz_abg.yaml


def register_layout(name: str, devices, admittance: bool = False, branch_b: bool = False) -> str:
    """Add a layout to LAYOUTS: `devices` = [(device name, 6-entry ctrlr_dof mask)] in TARGETS order -- any subset / order of the
    robot's devices, any row mask (what osc.py:134-138 stacks when a caller passes other targets, or a YAML sets another
    ctrlr_dof).  j_idx0 follows robot.py:50-55: rows of the robot's devices in ITS order, the devices that are not targeted
    keeping their YAML masks."""
    masks = dict(YAML_MASKS)
    masks.update({nm: list(m) for nm, m in devices})
    row0, r = {}, 0
    for nm in ROBOT_ORDER:
        row0[nm] = r
        r += int(sum(masks[nm]))
    LAYOUTS[name] = ([nm for nm, _ in devices], [list(map(bool, m)) for _, m in devices], [DEV_JOINTS[nm] for nm, _ in devices],
                     [row0[nm] for nm, _ in devices], dict(admittance=admittance, branch_b=branch_b))
    return name


def _m(bits: str):
    return [c == "1" for c in bits]


# Layouts a Dual-UR5 caller can reach beyond the four with a kernel instantiation of their own: target subsets, xyz-only arms,
# partial masks (tests/test_gpu_parity.py sweeps them on the KMAX-padded row16 kernels; tools/layout_sweep.py times them)
REACHABLE = [
    register_layout("r6", [("ur5right", _m("111111"))]),                                            # single arm          k 6, 1 dev
    register_layout("r3", [("ur5right", _m("111000"))]),                                            # single arm, xyz     k 3
    register_layout("b1", [("base", _m("000001"))]),                                                # stand yaw only      k 1
    register_layout("l2", [("ur5left", _m("110000"))]),                                             #                     k 2
    register_layout("br4", [("base", _m("000001")), ("ur5right", _m("111000"))]),                   # arm + base          k 4, 2 dev
    register_layout("rl5", [("ur5right", _m("111000")), ("ur5left", _m("110000"))]),                #                     k 5
    register_layout("br7", [("base", _m("000001")), ("ur5right", _m("111111"))]),                   # arm + base          k 7, 2 dev
    register_layout("rl8", [("ur5right", _m("111110")), ("ur5left", _m("111000"))]),                #                     k 8
    register_layout("rl9_admit", [("ur5right", _m("111111")), ("ur5left", _m("111000"))], admittance=True),    # k 9 + wrench
    register_layout("rlb10", [("ur5right", _m("111111")), ("ur5left", _m("111000")), ("base", _m("000001"))]),  # k 10, 3 dev
    register_layout("rlbr10", [("ur5right", _m("111000")), ("ur5left", _m("111000")), ("base", _m("000001")),
                               ("ur5right", _m("000111"))]),                                        # four target blocks  k 10, 4 dev
    register_layout("rlb11_branch_b", [("base", _m("000001")), ("ur5right", _m("111110")), ("ur5left", _m("111011"))],
                    branch_b=True),                                                                 # k 11, target velocities
    register_layout("brl14", [("base", _m("000011")), ("ur5right", _m("111111")), ("ur5left", _m("111111"))]),  # k 14
    register_layout("rlb16", [("ur5right", _m("111111")), ("ur5left", _m("111111")), ("base", _m("110011"))]),  # k 16 = IRLOSC_MAX_K
]

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
