"""The float64 rigid-body oracle (oracle/rigid_body.py) against MuJoCo-independent identities: MuJoCo is not available,
so what the front end must reproduce (M, EE Jacobians, bias forces, EE poses: robot.py:69, device.py:125-128,
osc.py:191 in the reference) is pinned by physics instead of by a recorded MuJoCo output."""
import numpy as np
import pytest

from oracle import rigid_body as rb


import os

MODELS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "irl_control_amd", "models")
# two trees through the same code: the Dual-UR5 (scenes/dual_ur5.xml: branching, 25 hinges) and the single arm of
# scenes/ur5.xml (a chain of 6 whose first body hangs on a world-fixed base)
PROBES = {"dual_ur5": ("ur_EE_ur5right", "ur_EE_ur5left", "ur_stand_dummy", "left_inner_finger_ur5left"),
          "ur5": ("EE", "link3", "link6")}


@pytest.fixture(scope="module", params=["dual_ur5", "ur5"])
def model(request):
    m = rb.Model(os.path.join(MODELS, request.param + ".json"))
    m.name = request.param
    return m


def _state(model, seed):
    rng = np.random.default_rng(seed)
    q = rng.uniform(-np.pi, np.pi, model.nj)
    for j, b in enumerate(model.joint_body):                 # gripper joints inside their ranges
        lo, hi = model.bodies[b]["joint"]["range"]
        if hi - lo < 3.0:
            q[j] = rng.uniform(lo, hi)
    return q, rng.normal(0.0, 0.5, model.nj)


def test_tree_matches_the_reference_scene(model):
    names = model.raw["joint_names"]
    if model.name == "ur5":                                   # scenes/ur5.xml:85-147: joint0..5 down one chain, motors on all six
        assert names == [f"joint{i}" for i in range(6)] and model.raw["actuator_joints"] == names
        assert list(np.nonzero(model.anc[model.body_id("EE")])[0]) == [0, 1, 2, 3, 4, 5]
        assert not model.anc[model.body_id("base_link")].any()
        return
    # joint / actuator numbering of SURVEY.md Appendix A (scenes/dual_ur5.xml)
    assert model.nj == 25 and names[0] == "ur_stand_joint" and names[1] == "joint0_ur5right" and names[13] == "joint0_ur5left"
    assert names[10] == "right_outer_knuckle_joint_ur5right" and names[22] == "right_outer_knuckle_joint_ur5left"
    acts = [names.index(a) for a in model.raw["actuator_joints"]]
    assert acts == [0, 1, 2, 3, 4, 5, 6, 10, 13, 14, 15, 16, 17, 18, 22]
    ee_r, ee_l = model.body_id("ur_EE_ur5right"), model.body_id("ur_EE_ur5left")
    assert list(np.nonzero(model.anc[ee_r])[0]) == [0, 1, 2, 3, 4, 5, 6]
    assert list(np.nonzero(model.anc[ee_l])[0]) == [0, 13, 14, 15, 16, 17, 18]
    assert list(np.nonzero(model.anc[model.body_id("ur_stand_dummy")])[0]) == [0]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_jacobians_equal_finite_differences_of_the_kinematics(model, seed):
    q, _ = _state(model, seed)
    kin = rb.kinematics(model, q)
    h = 1e-6
    for name in PROBES[model.name]:
        b = model.body_id(name)
        jp, jr = rb.body_jacobian(model, kin, b)
        for j in range(model.nj):
            dq = np.zeros(model.nj); dq[j] = h
            kp, km = rb.kinematics(model, q + dq), rb.kinematics(model, q - dq)
            assert np.allclose((kp["xpos"][b] - km["xpos"][b]) / (2 * h), jp[:, j], atol=1e-8)
            dR = (kp["xmat"][b] - km["xmat"][b]) / (2 * h) @ kin["xmat"][b].T          # [w]x
            w = np.array([dR[2, 1], dR[0, 2], dR[1, 0]])
            assert np.allclose(w, jr[:, j], atol=1e-8)


@pytest.mark.parametrize("seed", [0, 3])
def test_mass_matrix_equals_the_kinetic_energy_form(model, seed):
    q, qd = _state(model, seed)
    M, _, _ = rb.dynamics(model, q, qd)
    M2 = rb.mass_matrix_energy_form(model, q)
    assert np.allclose(M, M.T, atol=1e-14)
    assert np.allclose(M, M2, rtol=1e-12, atol=1e-13)
    assert np.linalg.eigvalsh(M).min() > 0
    if model.name == "dual_ur5":      # sparsity of the Dual-UR5 tree: the two arms only couple through the stand joint
        assert np.all(M[1:13, 13:25] == 0.0) and np.all(M[0, 1:] != 0.0)
    else:                             # a chain: every hinge is above or below every other one, no structural zero block
        assert np.count_nonzero(M) > 30


@pytest.mark.parametrize("seed", [0, 4])
def test_bias_forces_satisfy_lagranges_equations(model, seed):
    """bias_i = sum_jk (dM_ij/dq_k - 1/2 dM_jk/dq_i) qd_j qd_k + dPE/dq_i, derivatives by central differences."""
    q, qd = _state(model, seed)
    _, bias, _ = rb.dynamics(model, q, qd)
    h = 1e-5
    n = model.nj
    dM = np.zeros((n, n, n))
    dPE = np.zeros(n)
    for k in range(n):
        e = np.zeros(n); e[k] = h
        dM[:, :, k] = (rb.mass_matrix_energy_form(model, q + e) - rb.mass_matrix_energy_form(model, q - e)) / (2 * h)
        dPE[k] = (rb.potential_energy(model, q + e) - rb.potential_energy(model, q - e)) / (2 * h)
    c = np.einsum("ijk,j,k->i", dM, qd, qd) - 0.5 * np.einsum("jki,j,k->i", dM, qd, qd)
    assert np.allclose(bias, c + dPE, rtol=1e-6, atol=1e-6 * np.abs(bias).max())
    # gravity alone: zero velocity
    _, g, _ = rb.dynamics(model, q, np.zeros(n))
    assert np.allclose(g, dPE, rtol=1e-7, atol=1e-7 * np.abs(g).max())
    assert abs(g[0]) < 1e-9          # the first joint of both trees is vertical: gravity exerts no torque about it


def test_mesh_derived_inertia_of_the_base_links():
    """scenes/dual_ur5.xml:63,164: base_link_ur5right / left carry no <inertial>; MuJoCo integrates their mesh geom (link0.stl,
    density 1000).  tools/parse_mjcf.py did that integration in the build container; here the numbers it left in the model
    table are checked for plausibility against the mesh's bounding cylinder (r = 73.5 mm, h = 21.3 mm: 0.361 kg when solid)
    and for their effect: the yaw inertia of the stand joint grows by m (0.15^2 + ...) per link."""
    m = rb.Model(os.path.join(MODELS, "dual_ur5.json"))
    for name in ("base_link_ur5right", "base_link_ur5left"):
        b = m.bodies[m.body_id(name)]
        assert 0.25 < b["mass"] < 0.361 and b["inertia_from"].startswith("mesh geoms")
        assert abs(b["ipos"][0]) < 1e-3 and abs(b["ipos"][1]) < 1e-3 and 0.005 < b["ipos"][2] < 0.0213
        i1, i2, i3 = sorted(b["inertia"])
        assert i1 > 0 and i1 + i2 >= i3 * (1 - 1e-9)                                       # a physical inertia tensor
        assert abs(i3 - 0.5 * b["mass"] * 0.0735 ** 2) / i3 < 0.2                          # about the axis: ~ m r^2 / 2
        ex = b["mesh_inertia_exact"]
        assert 0.85 < ex["mass"] / b["mass"] < 1.0                                         # signed volumes: a little less
    q = np.zeros(m.nj)
    M_with = rb.mass_matrix_energy_form(m, q)[0, 0]
    bare = rb.Model(os.path.join(MODELS, "dual_ur5.json"))
    gain = 0.0
    for n in ("base_link_ur5right", "base_link_ur5left"):
        b = bare.bodies[bare.body_id(n)]
        gain += b["mass"] * 0.15 ** 2                       # the links sit 0.15 m off the yaw axis (dual_ur5.xml:63,164)
        b["mass"], b["inertia"] = 0.0, [0.0, 0.0, 0.0]
    M_without = rb.mass_matrix_energy_form(bare, q)[0, 0]
    assert gain < M_with - M_without < gain + 0.01 and (M_with - M_without) / M_with < 0.01


def test_records_have_the_abi_layout(model):
    if model.name != "dual_ur5":
        lay = dict(dev_names=["arm"], ctrlr_dof=[[True] * 6])
        r = rb.records(model, lay, dict(arm="EE"), *_state(model, 5))
        assert r["M"].shape == (6, 6) and r["J"].shape == (6, 6) and r["ee_pose"].shape == (1, 7)
        return
    q, qd = _state(model, 5)
    lay = dict(dev_names=["ur5right", "ur5left", "base"], ctrlr_dof=[[True] * 6, [True] * 6, [False] * 5 + [True]])
    ee = dict(base="ur_stand_dummy", ur5right="ur_EE_ur5right", ur5left="ur_EE_ur5left")
    r = rb.records(model, lay, ee, q, qd)
    assert r["M"].shape == (25, 25) and r["J"].shape == (13, 25) and r["ee_pose"].shape == (3, 7)
    assert np.allclose(r["J"][12], np.eye(25)[0])                     # base yaw row: rotation about z of joint 0 only
    assert np.all(r["J"][:6, 7:] == 0) and np.all(r["J"][6:12, 1:13] == 0)
    assert np.allclose(np.linalg.norm(r["ee_pose"][:, 3:], axis=1), 1.0)


def test_compiled_topology_header_matches_the_model_table():
    """csrc/topo_dual_ur5.hpp (the tree SHAPE the lane-per-robot front end is instantiated with) is what
    tools/gen_topology.py emits for models/dual_ur5.json today: nobody edited one without the other."""
    import io
    import os
    import sys
    from contextlib import redirect_stdout
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import gen_topology
    finally:
        sys.path.pop(0)
    buf = io.StringIO()
    cwd = os.getcwd()
    os.chdir(root)
    try:
        with redirect_stdout(buf):
            gen_topology.main("irl_control_amd/models/dual_ur5.json", "TopoDualUr5")
    finally:
        os.chdir(cwd)
    with open(os.path.join(root, "irl_control_amd", "csrc", "topo_dual_ur5.hpp")) as f:
        assert f.read() == buf.getvalue()


MJ_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mj_dual_ur5.npz")


def load_mujoco_fixture():
    """tests/golden/mj_dual_ur5.npz (oracle/make_mujoco_golden.py, official `mujoco` bindings) or a SKIP that says why."""
    if not os.path.exists(MJ_FIXTURE):
        pytest.skip("PARITY WITH MUJOCO UNPINNED: tests/golden/mj_dual_ur5.npz absent -- MuJoCo is in neither the build image nor the "
                    "GPU box; mint it with `python oracle/make_mujoco_golden.py <scenes/gain_test_scene.xml>` where `mujoco` is installed")
    return np.load(MJ_FIXTURE)


def test_against_mujoco_fixture():
    """Row f1 pinned to what the reference reads from MuJoCo (robot.py:69 mj_fullM, device.py:125-128 jacp / jacr, osc.py:191
    qfrc_bias, device.py:97-99 xpos / xquat): the rigid-body oracle -- and with it tools/parse_mjcf.py's reading of the MJCF
    (frame composition, inertiafromgeom for the mesh-only base links, armature, defaults) -- on MuJoCo's own states."""
    z = load_mujoco_fixture()
    m = rb.Model(os.path.join(MODELS, "dual_ur5.json"))
    assert [str(n) for n in z["joint_names"]] == m.raw["joint_names"]
    assert np.allclose(z["gravity"], m.gravity)
    dof, qadr = z["dofadr"], z["qposadr"]
    ee = [m.body_id(str(n)) for n in z["ee_bodies"]]
    worst = {}
    for i in range(z["qpos"].shape[0]):
        q, qd = z["qpos"][i][qadr], z["qvel"][i][dof]
        M, bias, kin = rb.dynamics(m, q, qd)
        got = {"fullM": M, "qfrc_bias": bias}
        want = {"fullM": z["fullM"][i][np.ix_(dof, dof)], "qfrc_bias": z["qfrc_bias"][i][dof]}
        for e, b in enumerate(ee):
            jp, jr = rb.body_jacobian(m, kin, b)
            got[f"jacp{e}"], want[f"jacp{e}"] = jp, z["jacp"][i, e][:, dof]
            got[f"jacr{e}"], want[f"jacr{e}"] = jr, z["jacr"][i, e][:, dof]
            got[f"xpos{e}"], want[f"xpos{e}"] = kin["xpos"][b], z["xpos"][i, e]
            sgn = np.sign(kin["xquat"][b] @ z["xquat"][i, e])                # q and -q are the same rotation
            got[f"xquat{e}"], want[f"xquat{e}"] = sgn * kin["xquat"][b], z["xquat"][i, e]
        for s_, nm in enumerate(z["ft_sites"]):
            site = m.site(str(nm))
            R = kin["xmat"][site["body"]] @ rb.quat2mat(np.array(site["quat"]))
            got[f"site{s_}"], want[f"site{s_}"] = R.reshape(9), z["site_xmat"][i, s_]
        for k in got:
            err = np.abs(got[k] - want[k]).max() / max(np.abs(want[k]).max(), 1e-12)
            worst[k] = max(worst.get(k, 0.0), err)
    print("oracle/rigid_body.py vs mujoco", str(z["mujoco_version"]), {k: f"{v:.1e}" for k, v in worst.items()})
    # kinematics and Jacobians are exact formulas; M and bias carry MuJoCo's own mesh-inertia rule for the two base links
    for k, v in worst.items():
        assert v <= (1e-9 if k.startswith(("jac", "xpos", "xquat", "site")) else 1e-6), (k, v)
