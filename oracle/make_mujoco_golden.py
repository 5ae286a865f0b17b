#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: mint tests/golden/mj_dual_ur5.npz with the OFFICIAL `mujoco` bindings -- the fixture that pins row f1
(oracle/rigid_body.py, the GPU front end, tools/parse_mjcf.py's reading of the MJCF) and row f2 (mujoco_backend.MujocoSim)
to what the reference actually reads from MuJoCo every tick:

    mj_fullM(qM)                      /root/reference/irl_control/robot.py:68-72
    mj_jacBody of the three EE bodies /root/reference/irl_control/device.py:115-133
    qfrc_bias                         /root/reference/irl_control/osc.py:190-191
    xpos / xquat of the EE bodies     /root/reference/irl_control/device.py:97-99
    site_xmat of the F/T frames       /root/reference/irl_control/device.py:135-170

CANNOT RUN IN THE BUILD IMAGE OR ON THE GPU BOX: neither has `mujoco` (nor mujoco_py).  Whoever has it:

    pip install mujoco
    python oracle/make_mujoco_golden.py /path/to/irl_control/scenes/gain_test_scene.xml      # default: /root/reference/...
    python -m pytest tests/test_rigid_body.py::test_against_mujoco_fixture tests/test_mujoco_backend.py -q

and commits tests/golden/mj_dual_ur5.npz (N = 64 states, ~0.5 MB).  Until then those tests SKIP with this reason, and
DESIGN.md keeps "parity with MuJoCo unpinned" for f1 / f2.

What is stored (all float64, batch-major): qpos[N, nq], qvel[N, nv] (hinges of the Dual-UR5 at random angles -- arm joints in
[-pi, pi], gripper joints inside their ranges -- free bodies of the scene left where they are), and MuJoCo's answers after
mj_forward: fullM[N, nv, nv], qfrc_bias[N, nv], jacp / jacr[N, 3, 3, nv] and xpos / xquat[N, 3, .] of ur_EE_ur5right,
ur_EE_ur5left, ur_stand_dummy, site_xmat[N, 2, 9] of ft_frame_ur5right / left, plus the index maps a consumer needs
(dof address and qpos address of every robot hinge in depth-first order, joint names, mujoco.__version__)."""
import os
import sys

import numpy as np

EE_BODIES = ["ur_EE_ur5right", "ur_EE_ur5left", "ur_stand_dummy"]
FT_SITES = ["ft_frame_ur5right", "ft_frame_ur5left"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "mj_dual_ur5.npz")


def main():
    try:
        import mujoco
    except ImportError:
        sys.exit("make_mujoco_golden.py needs the official `mujoco` package (pip install mujoco); it is in neither the build "
                 "image nor the GPU box -- see the header of this file")
    scene = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/irl_control/scenes/gain_test_scene.xml"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    m = mujoco.MjModel.from_xml_path(scene)
    d = mujoco.MjData(m)
    # the robot's hinges in MuJoCo's own (depth-first) order, starting at ur_stand_joint
    hinge = [j for j in range(m.njnt) if m.jnt_type[j] == mujoco.mjtJoint.mjJNT_HINGE]
    names = [mujoco.mj_id2name(m, mujoco.mjtObj.mjOBJ_JOINT, j) for j in hinge]
    assert names[0] == "ur_stand_joint" and len(hinge) == 25, names
    dofadr = np.array([m.jnt_dofadr[j] for j in hinge])
    qposadr = np.array([m.jnt_qposadr[j] for j in hinge])
    rng = np.random.default_rng(20241008)
    rec = {k: [] for k in ("qpos", "qvel", "fullM", "qfrc_bias", "jacp", "jacr", "xpos", "xquat", "site_xmat")}
    for _ in range(N):
        mujoco.mj_resetData(m, d)
        for j, qa, da in zip(hinge, qposadr, dofadr):
            lo, hi = m.jnt_range[j]
            narrow = bool(m.jnt_limited[j]) and (hi - lo) < 3.0
            d.qpos[qa] = rng.uniform(lo, hi) if narrow else rng.uniform(-np.pi, np.pi)
            d.qvel[da] = rng.normal(0.0, 0.5)
        mujoco.mj_forward(m, d)
        M = np.zeros((m.nv, m.nv))
        mujoco.mj_fullM(m, M, d.qM)
        jp, jr = np.zeros((3, 3, m.nv)), np.zeros((3, 3, m.nv))
        xp, xq = np.zeros((3, 3)), np.zeros((3, 4))
        for i, nm in enumerate(EE_BODIES):
            b = mujoco.mj_name2id(m, mujoco.mjtObj.mjOBJ_BODY, nm)
            mujoco.mj_jacBody(m, d, jp[i], jr[i], b)
            xp[i], xq[i] = d.xpos[b], d.xquat[b]
        sx = np.array([d.site_xmat[mujoco.mj_name2id(m, mujoco.mjtObj.mjOBJ_SITE, s)] for s in FT_SITES])
        for k, v in (("qpos", d.qpos), ("qvel", d.qvel), ("fullM", M), ("qfrc_bias", d.qfrc_bias), ("jacp", jp), ("jacr", jr),
                     ("xpos", xp), ("xquat", xq), ("site_xmat", sx)):
            rec[k].append(np.array(v, dtype=np.float64).copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(dofadr=dofadr, qposadr=qposadr, joint_names=np.array(names), ee_bodies=np.array(EE_BODIES), ft_sites=np.array(FT_SITES),
               mujoco_version=np.array(mujoco.__version__), scene=np.array(os.path.basename(scene)), gravity=np.array(m.opt.gravity))
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {N} states, nv = {m.nv}, mujoco {mujoco.__version__}")


if __name__ == "__main__":
    main()
