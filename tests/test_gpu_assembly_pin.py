"""Device-side state assembly pinned to the REFERENCE directly (SURVEY.md section 8 rows a13 / a14 batch form, f3).

The end-to-end fixtures (tests/golden/e2e_*.npz, minted by oracle/make_golden.py importing /root/reference) hold the RAW
simulator arrays the reference's Robot / Device read -- qM (mj_fullM), jacp / jacr, qvel, qfrc_bias, xpos / xquat, sensordata,
F/T site_xmat; nv = 25 and nv = 37 (two free bodies) -- next to what the reference made of them: its assembled M / J / dq
(robot.py:44-72, device.py:115-133) and the forces its OSC.generate returned.  Here those raw arrays go through
irlosc_upload_raw (host arrays) and irlosc_assemble_device (hipMalloc'ed arrays) and the step that follows is compared with the
REFERENCE's numbers -- not with the product's own host classes (tests/test_gpu_parity.py does that).  The host classes only
supply index tables (joint ids, actuator ids, YAML gains).
"""
import ctypes as C

import numpy as np
import pytest

from conftest import HipBuffers, app_from_e2e, live_mutation_phases, load_e2e
from irl_control_amd import BatchedOSC, OSCLayout, _lib, raw
from irl_control_amd.rigid_body import DUAL_UR5_EE
from irl_control_amd.robot import RobotState
from oracle import osc_oracle

pytestmark = pytest.mark.gpu
TOL64 = 1e-5


def raw_from_fixture(g, names, dtype=np.float64):
    """The fixture's raw arrays in irlosc_upload_raw's layouts, target devices `names` in targets order."""
    meta = g["meta"]
    B, nv = g["qM"].shape[0], g["qM"].shape[1]
    bi = [meta["ee_bodies"].index(DUAL_UR5_EE[nm]) for nm in names]
    xm = {"ur5right": g["xmat_right"], "ur5left": g["xmat_left"]}
    out = dict(qM=g["qM"], qvel=g["qvel"], qfrc_bias=g["qfrc_bias"],
               jacp=np.stack([g["jacp"][:, i].reshape(B, 3, nv) for i in bi], axis=1),
               jacr=np.stack([g["jacr"][:, i].reshape(B, 3, nv) for i in bi], axis=1),
               ee_xpos=np.stack([g["xpos"][:, i] for i in bi], axis=1),
               ee_xquat=np.stack([g["xquat"][:, i] for i in bi], axis=1),
               site_xmat=np.stack([xm[nm].reshape(B, 9) if nm in xm else np.zeros((B, 9)) for nm in names], axis=1),
               sensordata=g["sensordata"])
    return {k: np.ascontiguousarray(v, dtype=dtype) for k, v in out.items()}


def gains_of(osc_obj, robot, names):
    """YAML gain tables in BatchedOSC.set_gains form (what OSC.generate sends per tick, osc.py:19-39)."""
    cc = [osc_obj.device_configs[nm] for nm in names]
    return dict(kp=[c["kp"] for c in cc], kv=[c["kv"] for c in cc], ko=[c["ko"] for c in cc], k=[c["k"] for c in cc],
                d=[c["d"] for c in cc], max_vel=[robot.get_device(nm).max_vel or [0.0, 0.0] for nm in names],
                null_kv=(osc_obj.nullspace_config["kv"] if osc_obj.nullspace_config is not None else 0.0))


def reference_forces(robot, names, forces_flat):
    """The reference's per-device force vectors scattered back to joint positions: -> (u_ref [B, n] with NaN where the
    reference returned nothing, mask)."""
    B = forces_flat.shape[0]
    u = np.full((B, robot.num_joints_total), np.nan)
    off = 0
    for nm in names:
        trn = list(robot.sub_devices_dict[nm].actuator_trnids)
        u[:, trn] = forces_flat[:, off:off + len(trn)]
        off += len(trn)
    assert off == forces_flat.shape[1]
    return u


def rel_err(u, ref):
    m = ~np.isnan(ref)
    d = np.where(m, np.abs(u - np.where(m, ref, 0.0)), 0.0)
    return d.max(axis=1) / np.nanmax(np.abs(ref), axis=1)


def assemble_device(osc, hb, desc, arrs, B, slot=0):
    dev = {k: hb.to_device(v) for k, v in arrs.items()}
    rc = osc.lib.irlosc_assemble_device(osc._h, slot, B, C.byref(desc), dev["qM"], dev["qvel"], dev["qfrc_bias"], dev["jacp"],
                                        dev["jacr"], dev["ee_xpos"], dev["ee_xquat"], dev["site_xmat"], dev["sensordata"], None)
    assert rc == 0, osc.lib.irlosc_last_error(osc._h)
    osc._B[slot] = B


@pytest.mark.parametrize("entry", ["upload_raw", "assemble_device"])
@pytest.mark.parametrize("name", ["e2e_gain_test", "e2e_admit_test", "e2e_single_arm"])
def test_raw_arrays_through_device_assembly_give_the_references_forces(name, entry):
    """fixture raw arrays (nv = 25: gain_test, k = 7; nv = 37: admit_test, k = 12 + wrench; nv = 31: one arm alone, k = 6 on the padded
    kernel) -> assembly ON THE GPU -> step ->
    the reference's forces, fp64 <= 1e-5 (measured ~1e-12); the assembled records equal the reference's assembled M / J / dq
    bit for bit; flags equal the reference's branch."""
    g = load_e2e(name)
    meta = g["meta"]
    names = meta["target_order"]
    B = g["qM"].shape[0]
    app, robot, osc_obj, _ = app_from_e2e(g, 0)
    devs = [robot.get_device(nm) for nm in names]
    _, J_idxs = robot.get_state(RobotState.J)
    lay = OSCLayout.from_devices(devs, robot, use_g=meta["use_g"], admittance=meta["admittance"], nullspace=meta["nullspace"],
                                 J_idxs=J_idxs)
    arrs = raw_from_fixture(g, names)
    desc = raw.raw_desc(robot, names, arrs["sensordata"].shape[1])
    tgt = np.concatenate([g["tgt_xyz"], g["tgt_quat"]], axis=2)
    osc = BatchedOSC(lay, B, dtype=np.float64)
    osc.set_gains(**gains_of(osc_obj, robot, names))
    hb = HipBuffers()
    if entry == "upload_raw":
        osc.upload_raw(desc, **arrs)
    else:
        assemble_device(osc, hb, desc, arrs, B)
    rec = osc.download_records()
    assert np.array_equal(rec["M"], g["M"]) and np.array_equal(rec["J"], g["J"]) and np.array_equal(rec["dq"], g["dq"])
    osc.set_targets(tgt)
    u, fl = osc.step(return_flags=True)
    osc.close()
    hb.free()
    ref = reference_forces(robot, names, g["forces_flat"])
    err = rel_err(u, ref)
    assert err.max() <= TOL64, (name, entry, err)
    assert err.max() <= 1e-9, (name, entry, err)                # what fp64 actually delivers on these states
    assert not np.any(fl & (_lib.FLAG_NONFINITE | _lib.FLAG_M_NOT_PD | _lib.FLAG_BAD_JIDX))
    dets = np.array([osc_oracle.task_inertia(g["J"][b], g["M"][b])[3] for b in range(B)])
    assert np.array_equal((fl & _lib.FLAG_PINV_BRANCH) != 0, np.abs(dets) < 1e-4)


@pytest.mark.parametrize("name", ["e2e_gain_test", "e2e_admit_test"])
def test_raw_arrays_as_float32_records_on_the_mixed_path(name):
    """The same raw arrays as float32 (what an fp32 simulator hands over): assembly on the GPU into float32 records, fp64
    arithmetic (row16 mixed path, or the generic kernel for layouts without one).  Reference for the comparison = the oracle
    (pinned to the reference <= 1e-9) on the REFERENCE's assembled records rounded to float32 -- the assembly is pure picking,
    so rounding commutes with it; the F/T rotation runs in float32 on float32 records and gets eps32-level slack."""
    g = load_e2e(name)
    meta = g["meta"]
    names = meta["target_order"]
    B = g["qM"].shape[0]
    app, robot, osc_obj, _ = app_from_e2e(g, 0)
    devs = [robot.get_device(nm) for nm in names]
    _, J_idxs = robot.get_state(RobotState.J)
    lay = OSCLayout.from_devices(devs, robot, use_g=meta["use_g"], admittance=meta["admittance"], nullspace=meta["nullspace"],
                                 J_idxs=J_idxs)
    gains = gains_of(osc_obj, robot, names)
    arrs = raw_from_fixture(g, names, dtype=np.float32)
    desc = raw.raw_desc(robot, names, arrs["sensordata"].shape[1])
    r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    tgt = r32(np.concatenate([g["tgt_xyz"], g["tgt_quat"]], axis=2))
    osc = BatchedOSC(lay, B, dtype=np.float32)
    osc.set_gains(**gains)
    osc.upload_raw(desc, **arrs)
    rec = osc.download_records()
    assert np.array_equal(rec["M"], g["M"].astype(np.float32)) and np.array_equal(rec["J"], g["J"].astype(np.float32))
    osc.set_targets(tgt)
    u = osc.step().astype(np.float64)
    osc.close()
    ids = list(robot.joint_ids_all)
    bias = r32(g["qfrc_bias"])[:, ids]
    ee = np.concatenate([arrs["ee_xpos"], arrs["ee_xquat"]], axis=2).astype(np.float64)
    wr = None
    if meta["admittance"]:                   # R(site) @ sensordata[slice] in float32, index order (device.py:139-170)
        wr = np.zeros((B, len(names), 6), np.float32)
        for i, nm in enumerate(names):
            R = arrs["site_xmat"][:, i].reshape(B, 3, 3)
            f0, t0 = desc.ft_force0[i], desc.ft_torque0[i]
            for part, s0 in enumerate((f0, t0)):
                s = arrs["sensordata"][:, s0:s0 + 3]
                wr[:, i, 3 * part:3 * part + 3] = (R[:, :, 0] * s[:, :1] + R[:, :, 1] * s[:, 1:2]) + R[:, :, 2] * s[:, 2:3]
        wr = wr.astype(np.float64)
    ref = osc_oracle.generate_batch(lay.as_oracle_dict(), gains, r32(g["M"]), r32(g["J"]), r32(g["dq"]), bias, ee, tgt, wr)
    m = np.zeros(lay.n, bool)
    for nm in names:
        m[list(robot.sub_devices_dict[nm].actuator_trnids)] = True
    err = np.abs(u - ref)[:, m].max(axis=1) / np.abs(ref)[:, m].max(axis=1)
    assert err.max() <= 2e-6, (name, err)          # float32 OUTPUT rounding (6e-8) x a few; arithmetic is fp64


def test_live_mutation_phases_through_device_assembly():
    """e2e_live_mutations (ps_move_example.py:137-150 / insertion_task.py:294: ctrlr_dof_abg and max_vel[0] changed between ticks on
    live devices): per phase, the fixture's raw arrays -> irlosc_upload_raw on a context built for the phase's layout ->
    the reference's forces of that phase."""
    g = load_e2e("e2e_live_mutations")
    meta = g["meta"]
    names = meta["target_order"]
    B = g["qM"].shape[0]
    arrs = raw_from_fixture(g, names)
    seen = 0
    for p, app, robot, osc_obj, _ in live_mutation_phases(g, 0):
        devs = [robot.get_device(nm) for nm in names]
        _, J_idxs = robot.get_state(RobotState.J)
        lay = OSCLayout.from_devices(devs, robot, use_g=meta["use_g"], admittance=meta["admittance"],
                                     nullspace=meta["nullspace"], J_idxs=J_idxs)
        desc = raw.raw_desc(robot, names, arrs["sensordata"].shape[1])
        osc = BatchedOSC(lay, B, dtype=np.float64)
        osc.set_gains(**gains_of(osc_obj, robot, names))
        osc.upload_raw(desc, **arrs)
        osc.set_targets(np.concatenate([g["tgt_xyz"][:, p], g["tgt_quat"][:, p]], axis=2))
        u = osc.step()
        osc.close()
        err = rel_err(u, reference_forces(robot, names, g["forces_flat"][:, p]))
        assert err.max() <= 1e-9, (p, err)
        seen += 1
    assert seen == len(meta["phases"])


def _tiled_admit_states(B, seed):
    """65 536 robots out of the six states of e2e_admit_test (nv = 37): per-instance perturbations that keep M symmetric
    positive definite and the zero pattern of every array."""
    g = load_e2e("e2e_admit_test")
    names = g["meta"]["target_order"]
    a6 = raw_from_fixture(g, names)
    rng = np.random.default_rng(seed)
    rep = lambda v: np.tile(v, (B // v.shape[0] + 1,) + (1,) * (v.ndim - 1))[:B].copy()
    a = {k: rep(v) for k, v in a6.items()}
    nv = a["qM"].shape[1]
    s = 1.0 + 0.2 * rng.random((B, nv))                      # D M D with D = diag(s): symmetric, positive definite, same zeros
    a["qM"] *= s[:, :, None] * s[:, None, :]
    for k in ("qvel", "qfrc_bias", "jacp", "jacr", "sensordata"):
        a[k] *= 1.0 + 0.1 * rng.standard_normal(a[k].shape)
    a["ee_xpos"] += 0.05 * rng.standard_normal(a["ee_xpos"].shape)
    tgt = rep(np.concatenate([g["tgt_xyz"], g["tgt_quat"]], axis=2))
    tgt[:, :, :3] += 0.1 * rng.standard_normal(tgt[:, :, :3].shape)
    return g, names, a, tgt


def test_upload_raw_full_size_properties_nv37():
    """irlosc_upload_raw at the headline batch size, nv = 37 > n = 25 (the row / column picking really picks), k = 12 with the
    admittance wrench: (1) a stratified sample of 1 024 robots against the oracle fed with records assembled on the host by
    the reference's formulas (robot.py:44-72, device.py:115-170 restated in NumPy right here) <= 1e-5; (2) the first half of the
    batch alone gives the same bits; (3) torques are affine in sensordata (the wrench enters through R(site) only)."""
    B = 65536
    g, names, a, tgt = _tiled_admit_states(B, seed=20241030)
    meta = g["meta"]
    app, robot, osc_obj, _ = app_from_e2e(g, 0)
    devs = [robot.get_device(nm) for nm in names]
    _, J_idxs = robot.get_state(RobotState.J)
    lay = OSCLayout.from_devices(devs, robot, use_g=True, admittance=True, nullspace=True, J_idxs=J_idxs)
    gains = gains_of(osc_obj, robot, names)
    desc = raw.raw_desc(robot, names, a["sensordata"].shape[1])
    osc = BatchedOSC(lay, B, dtype=np.float64)
    assert "row16" in osc.kernel_name
    osc.set_gains(**gains)
    osc.upload_raw(desc, **a)
    osc.set_targets(tgt)
    u, fl = osc.step(return_flags=True)
    assert np.all(np.isfinite(u)) and not np.any(fl & (_lib.FLAG_NONFINITE | _lib.FLAG_M_NOT_PD))
    # (1) host assembly by the reference's formulas on a sample, then the oracle
    idx = np.arange(7, B, 64)
    ids = np.array(robot.joint_ids_all)
    dq_src = np.array([desc.dq_src[p] for p in range(lay.n)])
    M = a["qM"][idx][:, ids][:, :, ids]
    dq = np.where(dq_src >= 0, a["qvel"][idx][:, np.maximum(dq_src, 0)], 0.0)
    bias = a["qfrc_bias"][idx][:, ids]
    Jrows, wr = [], np.zeros((len(idx), len(names), 6))
    for i, dv in enumerate(devs):
        J6 = np.concatenate([a["jacp"][idx, i], a["jacr"][idx, i]], axis=1)[:, :, ids]
        Jrows.append(J6[:, np.array(dv.ctrlr_dof, bool)])
        R = a["site_xmat"][idx, i].reshape(-1, 3, 3)
        for part, s0 in enumerate((desc.ft_force0[i], desc.ft_torque0[i])):
            wr[:, i, 3 * part:3 * part + 3] = np.einsum("bij,bj->bi", R, a["sensordata"][idx, s0:s0 + 3])
    J = np.concatenate(Jrows, axis=1)
    ee = np.concatenate([a["ee_xpos"][idx], a["ee_xquat"][idx]], axis=2)
    ref = osc_oracle.generate_batch(lay.as_oracle_dict(), gains, M, J, dq, bias, ee, tgt[idx], wr)
    err = np.abs(u[idx] - ref).max(axis=1) / np.abs(ref).max(axis=1)
    assert err.max() <= TOL64, err.max()
    # (2) half the batch, same bits
    osc.upload_raw(desc, **{k: v[:B // 2] for k, v in a.items()})
    osc.set_targets(tgt[:B // 2])
    assert np.array_equal(osc.step(), u[:B // 2])
    # (3) affine in sensordata
    rng = np.random.default_rng(3)
    s1 = a["sensordata"] + rng.standard_normal(a["sensordata"].shape)
    outs = []
    for sd in (s1, 0.5 * (a["sensordata"] + s1)):
        osc.upload_raw(desc, **dict(a, sensordata=sd))
        osc.set_targets(tgt)
        outs.append(osc.step())
    osc.close()
    mid = 0.5 * (u + outs[0])
    scale = np.maximum(np.abs(u).max(axis=1), np.abs(outs[0]).max(axis=1))
    assert (np.abs(outs[1] - mid).max(axis=1) / scale).max() <= 1e-9


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("free_dofs", [0, 12])
def test_qM_in_mujocos_sparse_form_gives_the_records_of_the_dense_form(dtype, free_dofs):
    """irlosc_upload_raw_sparse: mjData.qM as MuJoCo holds it (nM entries per robot: every dof's run up the tree through dof_parentid,
    starting at dof_Madr) expanded ON THE GPU -- what robot.py:68-72 has mj_fullM do on the host for every robot and tick.  Physical M
    of random Dual-UR5 states (tree pattern), optionally behind `free_dofs` dofs of free bodies in front of the robot's (the
    admit_test scene: nv = 37): the slot's records and the step's torques equal those of the dense raw upload bit for bit."""
    from irl_control_amd import synth
    from irl_control_amd.rigid_body import RigidBodyModel
    B, n = 300, 25
    lay = synth.make_layout("k13")
    model = RigidBodyModel.load("dual_ur5")
    rng = np.random.default_rng(11)
    fe = BatchedOSC(lay, B, dtype=np.float64)
    fe.set_model(model)
    fe.upload_q(*model.random_state(rng, B))
    fe.frontend()
    rec = fe.download_records(0)
    fe.close()
    _, gains, g = synth.make_batch("k13", B, seed=12, dtype=dtype)
    nv, f = n + free_dofs, free_dofs
    # the robot's dofs come after the free bodies' (MuJoCo numbers dofs in body order: world.xml's free bodies first)
    qM = np.zeros((B, nv, nv))
    qM[:, f:, f:] = rec["M"]
    for k in range(f):
        qM[:, k, k] = 1.0 + k
    PARENT = [-1, 0, 1, 2, 3, 4, 5, 6, 7, 6, 6, 10, 6, 0, 13, 14, 15, 16, 17, 18, 19, 18, 18, 22, 18]      # hinge tree
    par = [(k - 1 if k % 6 else -1) for k in range(f)] + [(p + f if p >= 0 else -1) for p in PARENT]        # a free body: a chain of six dofs
    d = _lib.RawDesc()
    d.nv, d.n_sensor = nv, 0
    for p_ in range(32):
        d.joint_ids[p_] = p_ + f if p_ < n else 0
        d.dq_src[p_] = p_ + f if p_ < n else -1
    for i in range(4):
        d.ft_force0[i] = d.ft_torque0[i] = -1
    jacp, jacr = np.zeros((B, 3, 3, nv)), np.zeros((B, 3, 3, nv))
    jacp[:, 0, :, f:], jacr[:, 0, :, f:] = rec["J"][:, 0:3], rec["J"][:, 3:6]
    jacp[:, 1, :, f:], jacr[:, 1, :, f:] = rec["J"][:, 6:9], rec["J"][:, 9:12]
    jacr[:, 2, 2, f:] = rec["J"][:, 12]
    qvel, qb = np.zeros((B, nv)), np.zeros((B, nv))
    qvel[:, f:], qb[:, f:] = rec["dq"], rec["bias"]
    common = dict(qvel=qvel, qfrc_bias=qb, jacp=jacp, jacr=jacr, ee_xpos=np.ascontiguousarray(rec["ee_pose"][:, :, :3]),
                  ee_xquat=np.ascontiguousarray(rec["ee_pose"][:, :, 3:]))
    common = {k: v.astype(dtype) for k, v in common.items()}
    ql = raw.qm_layout(par)
    def run_len(i):
        ln = 0
        while i >= 0:
            ln, i = ln + 1, par[i]
        return ln
    assert ql.nM == sum(run_len(i) for i in range(nv)) == 155 + 21 * (f // 6)      # the Dual-UR5's 155 + 21 per free body
    sparse = raw.pack_qM(qM, ql).astype(dtype)
    assert sparse.shape[1] == ql.nM and ql.nM < nv * nv / 3
    osc = BatchedOSC(lay, B, dtype=dtype, n_slots=2)
    osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
    osc.upload_raw(d, qM.astype(dtype), slot=0, **common)
    osc.upload_raw(d, sparse, slot=1, qm_layout=ql, **common)
    r0, r1 = osc.download_records(0), osc.download_records(1)
    for k in ("M", "J", "dq", "bias", "ee_pose"):
        assert np.array_equal(r0[k], r1[k]), k
    assert np.array_equal(r1["M"].astype(np.float64), rec["M"].astype(dtype).astype(np.float64))
    assert osc.slot_structure(0) and osc.slot_structure(1)
    for s in (0, 1):
        osc.set_targets(g["tgt_pose"], slot=s)
    u0, f0 = osc.step(slot=0, return_flags=True)
    u1, f1 = osc.step(slot=1, return_flags=True)
    assert np.array_equal(u0, u1) and np.array_equal(f0, f1) and np.all(np.isfinite(u1))
    # a layout that runs past nM, or a child before its parent, is refused
    bad = raw.qm_layout(par)
    bad.nM = ql.nM - 1
    with pytest.raises(_lib.IrloscError, match="exceeds nM"):
        osc.upload_raw(d, sparse[:, :-1].copy(), slot=1, qm_layout=bad, **common)
    osc.close()
