"""Every layout a Dual-UR5 caller can reach runs on a row16-class kernel (no 26x cliff to the generic kernel).

The reference stacks the Jacobian over WHATEVER targets it is handed (osc.py:134-138) with whatever row masks the YAML / a live
re-mask gives the devices (device.py:36, examples/ps_move_example.py:137-150): single arm, arm + base, xyz-only arms, partial
masks.  The four shapes of the shipped examples have kernel instantiations of their own; every other n = 25 layout goes to the
KMAX-padded variant of the same kernel (csrc/osc_row16.hpp, PAD; tiers 4 / 7 / 10 / 13 / 16).  This file sweeps them against the
oracle -- synthetic dense records, physical records in the tree form, the fused path from joint coordinates -- and checks that the
padding changes no bit (a padded tier against the exact instantiation of the same shape).
"""
import numpy as np
import pytest

from irl_control_amd import BatchedOSC, _lib, synth
from oracle import osc_oracle
from conftest import oracle_on_all
from test_gpu_parity import _from_q_setup, in_parity_domain, rel_err, run_gpu

pytestmark = pytest.mark.gpu
TOL64 = 1e-5


def _oracle(lay, gains, g, idx):
    return osc_oracle.generate_batch(lay.as_oracle_dict(), gains, g["M"], g["J"], g["dq"], g["bias"], g["ee_pose"], g["tgt_pose"],
                                     g.get("wrench"), g.get("tgt_vel"), idx=idx)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cfg", synth.REACHABLE)
def test_every_reachable_layout_runs_on_a_row16_kernel_and_matches_the_oracle(cfg, dtype):
    """4 096 synthetic instances per layout (dense records, no tree zeros): KERNEL_AUTO picks a row16-class kernel, torques within
    1e-5 of the oracle on the parity domain, the PINV flag equals the reference's det test (osc.py:52), a ragged sub-batch gives
    the same bits."""
    B = 4096
    lay, gains, g = synth.make_batch(cfg, B, seed=20241008 + lay_seed(cfg), dtype=dtype)
    osc = BatchedOSC(lay, B, dtype=dtype)
    assert "row16" in osc.kernel_name, osc.kernel_name
    assert osc.kernel_class in (_lib.CLASS_ROW16, _lib.CLASS_ROW16_PADDED)
    if (lay.k, lay.ndev) not in ((13, 3), (12, 2), (7, 3), (6, 2)):
        assert osc.kernel_class == _lib.CLASS_ROW16_PADDED and f"_pad{min(t for t in (4, 7, 10, 13, 16) if t >= lay.k)}" in osc.kernel_name
    osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
    u, fl = osc.generate_batched(g["M"], g["J"], g["dq"], g["bias"], g["ee_pose"], g["tgt_pose"], g.get("tgt_vel"), g.get("wrench"),
                                 return_flags=True)
    u = u.astype(np.float64)
    g64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in g.items()}
    idx = np.arange(B)                                   # the oracle on EVERY instance (forked over the host cores)
    ref, dom, _, _, det = oracle_on_all(lay.as_oracle_dict(), gains, g64)
    assert dom.mean() > 0.8, dom.mean()
    err = rel_err(u[idx], ref[idx])
    tol = TOL64 if dtype == np.float64 else 2e-5        # float32 OUTPUT words: a large component next to a small one
    assert err[dom].max() <= tol, (cfg, osc.kernel_name, float(err[dom].max()))
    assert not np.any(fl[idx][dom] & (_lib.FLAG_NONFINITE | _lib.FLAG_M_NOT_PD))
    clear = np.abs(np.abs(det) / 1e-4 - 1.0) > 1e-6
    assert np.array_equal((fl[idx][dom & clear] & _lib.FLAG_PINV_BRANCH) != 0, np.abs(det[dom & clear]) < 1e-4)
    if "branch_b" in cfg:
        assert (fl & _lib.FLAG_VEL_BRANCH_B).astype(bool).mean() > 0.5
    # ragged sub-batch (not a multiple of the 4 instances of a wave), same bits
    n2 = 1001
    u2 = osc.generate_batched(*(g[k][:n2] for k in ("M", "J", "dq", "bias", "ee_pose", "tgt_pose")),
                              None if g.get("tgt_vel") is None else g["tgt_vel"][:n2], None if g.get("wrench") is None else g["wrench"][:n2])
    osc.close()
    assert np.array_equal(u2.astype(np.float64), u[:n2])


def lay_seed(cfg):
    return 1 + synth.REACHABLE.index(cfg)


@pytest.mark.parametrize("cfg", synth.REACHABLE)
def test_every_reachable_layout_on_physical_states_tree_form_and_fused(cfg):
    """Physical Dual-UR5 states (random joint coordinates, every 9th robot with stretched / folded arms): the fused path from
    (qpos, qvel) and the path through dense records in the tree-structured form run the padded kernels too (FROMQ / TopoDualUr5
    instantiations) and agree to rounding with equal flags; the dense path is within 1e-5 of the oracle on the records the front
    end wrote."""
    B = 1024 + 13
    lay, gains, g, model, osc, states = _from_q_setup(cfg, B, np.float64, seed=300 + lay_seed(cfg), singular_every=9)
    assert "fused" in osc.from_q_name
    u_f, fl_f = osc.step_q(return_flags=True)
    osc.frontend()
    assert osc.slot_structure(0)
    rec = osc.download_records(0)
    u_d, fl_d = osc.step(return_flags=True)
    osc.close()
    assert np.all(np.isfinite(u_f)) and np.array_equal(fl_f, fl_d)
    d = np.abs(u_f - u_d).max(axis=1) / np.abs(u_d).max(axis=1)
    assert d.max() <= 1e-8, float(d.max())
    # (no F/T reading was ever uploaded into the slot: the wrench term of an admittance layout is zero on this path)
    r = dict(rec, tgt_pose=g["tgt_pose"], wrench=None if g.get("wrench") is None else np.zeros_like(g["wrench"]), tgt_vel=g.get("tgt_vel"))
    idx = np.arange(0, B, 4)
    ref = _oracle(lay, gains, r, idx)
    dom = np.array([in_parity_domain(*osc_oracle.task_inertia(rec["J"][b], rec["M"][b])[2:]) for b in idx])
    assert dom.mean() > 0.5
    assert rel_err(u_d[idx], ref[idx])[dom].max() <= TOL64


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cfg", ["k13", "k12_admit", "k7", "k6", "k13_branch_b"])
def test_padding_changes_no_bit(cfg, dtype, monkeypatch):
    """The KMAX-padded variant against the exact instantiation of the same shape (IRLOSC_FORCE_PAD=1 sends a shape that has its own
    kernel to the padded tier: k 13 -> 13, 12 -> 13, 7 -> 7, 6 -> 7): padding only adds exact zeros, so torques and flags are
    identical -- det, trace, the eigen stage and the 1e-5 cut see the k x k problem.  Stress batch: eigenvalues all around the
    cut, so the eigen stage runs on a large share of the instances; and the fused path on physical states."""
    B = 4096 + 3
    lay, gains, g = synth.make_batch(cfg, B, seed=77, dtype=dtype)
    u_e, fl_e, name_e = run_gpu(lay, gains, g, dtype)
    lay2, gains2, g2, model, osc, _ = _from_q_setup(cfg, 1024 + 5, dtype, seed=78, singular_every=5)
    uq_e, flq_e = osc.step_q(return_flags=True)
    osc.close()
    monkeypatch.setenv("IRLOSC_FORCE_PAD", "1")
    u_p, fl_p, name_p = run_gpu(lay, gains, g, dtype)
    lay2, gains2, g2, model, osc, _ = _from_q_setup(cfg, 1024 + 5, dtype, seed=78, singular_every=5)
    assert "pad" in osc.kernel_name
    uq_p, flq_p = osc.step_q(return_flags=True)
    osc.close()
    assert "pad" in name_p and "pad" not in name_e
    assert np.array_equal(fl_e, fl_p) and np.array_equal(u_e, u_p, equal_nan=True)
    assert (flq_e & _lib.FLAG_EIGEN_PATH).astype(bool).mean() > 0.005 or cfg in ("k7", "k6")      # the eigen stage ran on both
    assert np.array_equal(flq_e, flq_p) and np.array_equal(uq_e, uq_p, equal_nan=True)


def test_kernel_class_reports_the_generic_fallback():
    """irlosc_kernel_class: a C caller that asked for AUTO can see what it got.  n != 25 is the one case left on the generic
    kernel (BatchedOSC warns at throughput batch sizes); an explicit KERNEL_GENERIC reports itself too."""
    from irl_control_amd.layout import OSCLayout as L
    lay = L(n=12, dev_names=["arm"], ctrlr_dof=[[True] * 6], joint_ids=[list(range(0, 12))], j_idx0=[0])
    with pytest.warns(RuntimeWarning, match="no throughput kernel"):
        osc = BatchedOSC(lay, 2048)
    assert osc.kernel_class == _lib.CLASS_GENERIC and "generic" in osc.kernel_name
    osc.close()
    osc = BatchedOSC(synth.make_layout("r6"), 64, kernel=_lib.KERNEL_GENERIC)
    assert osc.kernel_class == _lib.CLASS_GENERIC
    osc.close()
    osc = BatchedOSC(synth.make_layout("k13"), 64)
    assert osc.kernel_class == _lib.CLASS_ROW16 and osc.kernel_name.endswith("k13")
    osc.close()
    osc = BatchedOSC(synth.make_layout("r6"), 64)
    assert osc.kernel_class == _lib.CLASS_ROW16_PADDED and osc.kernel_name.endswith("k6_ndev1_pad7")
    osc.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cfg", ["k13", "k12_admit", "k13_branch_b", "k7", "r6", "rlbr10", "rlb16"])
def test_task_pass_changes_no_bit(cfg, dtype, monkeypatch):
    """Dense records on the row16 path: part 1 of the task-space signal (osc.py:101-118,70-99,160-168) runs as a pass of its own ahead of
    the kernel (osc_task_rows_dense_kernel: one lane per (instance, device)); IRLOSC_TASK_PASS=0 keeps it inside the kernel (sixteen
    lanes per instance).  Same formulas in the same order: torques and flags identical, on exact and padded kernels, with the wrench
    and the target-velocity branch, on a ragged batch, float64 and float32 records."""
    B = 4096 + 37
    lay, gains, g = synth.make_batch(cfg, B, seed=91, dtype=dtype)
    u_p, fl_p, name = run_gpu(lay, gains, g, dtype, kernel=_lib.KERNEL_ROW16)
    monkeypatch.setenv("IRLOSC_TASK_PASS", "0")
    u_k, fl_k, name_k = run_gpu(lay, gains, g, dtype, kernel=_lib.KERNEL_ROW16)
    assert "row16" in name and name == name_k
    assert np.array_equal(fl_p, fl_k) and np.array_equal(u_p, u_k, equal_nan=True)
    assert np.isfinite(u_p).all() and np.abs(u_p).max() > 0
    if "branch_b" in cfg:
        assert (fl_p & _lib.FLAG_VEL_BRANCH_B).astype(bool).mean() > 0.5
