#!/usr/bin/env python3
"""Throughput of EVERY layout a Dual-UR5 caller can reach (synth.REACHABLE + the four with kernel instantiations of their own)
at the headline batch size: dense records of physical states (tree form), synthetic dense records, and the fused path from joint
coordinates; plus the cost of the KMAX padding itself (IRLOSC_FORCE_PAD=1 sends a shape with its own kernel to its padded tier).

    python tools/layout_sweep.py [--batch 65536] [--steps 400] [--out profiles/r05_layout_sweep.json]

One line per layout: k, ndev, kernel name, steps/s and fraction of the 8 TB/s HBM peak on the SURVEY.md 8(d) algorithmic bytes,
worst error against the oracle on a sample (the checker only).  Before round 5 every layout outside the four instantiated shapes ran on
osc_generic at 2.2e7 steps/s."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irl_control_amd import BatchedOSC, _lib, synth                 # noqa: E402
from irl_control_amd.rigid_body import RigidBodyModel              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--slots", type=int, default=16)      # (two trains are in flight: fewer slots let the second reader of a batch hit the Infinity Cache, NOTES round 6)
ap.add_argument("--oracle", type=int, default=256)
ap.add_argument("--layouts", default="")
ap.add_argument("--dtype", default="f64")
ap.add_argument("--out", default="")
a = ap.parse_args()
dt = np.float64 if a.dtype == "f64" else np.float32
esz = np.dtype(dt).itemsize
B = a.batch
model = RigidBodyModel.load("dual_ur5")
EXACT = ["k13", "k12_admit", "k7", "k6"]
names = a.layouts.split(",") if a.layouts else EXACT + list(synth.REACHABLE)


def alg_bytes(lay):
    return esz * (lay.n * lay.n + lay.k * lay.n + 2 * lay.n + 14 * lay.ndev + (6 * lay.ndev if lay.admittance else 0) + lay.n)


def oracle_err(lay, gains, rec, tgt, wr, tv, u):
    from oracle import osc_oracle                                   # the checker only
    n = min(a.oracle, B)
    r = {k: np.asarray(v[:n], dtype=np.float64) for k, v in rec.items()}
    ref = osc_oracle.generate_batch(lay.as_oracle_dict(), gains, r["M"], r["J"], r["dq"], r["bias"], r["ee_pose"],
                                    np.asarray(tgt[:n], np.float64), None if wr is None else np.asarray(wr[:n], np.float64),
                                    None if tv is None else np.asarray(tv[:n], np.float64))
    err = np.abs(u[:n].astype(np.float64) - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1e-300)
    dom = []
    for b in range(n):
        _, _, Mxi, det = osc_oracle.task_inertia(r["J"][b], r["M"][b])
        s = np.linalg.svd(Mxi, compute_uv=False)
        dom.append(s[-1] > 1e-12 * s[0] if abs(det) >= 1e-4 else not np.any(np.abs(s / s[0] / 1e-5 - 1.0) < 1e-2))
    dom = np.array(dom)
    return float(err[dom].max()) if dom.any() else None, int((err[dom] > 1e-5).sum())


def run(cfg, force_pad=False):
    os.environ["IRLOSC_FORCE_PAD"] = "1" if force_pad else "0"
    lay = synth.make_layout(cfg)
    _, gains, arr = synth.make_batch(cfg, B, seed=7, dtype=dt)
    rng = np.random.default_rng(5)
    osc = BatchedOSC(lay, B, dtype=dt, n_slots=a.slots)
    osc.set_model(model)
    osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
    out = dict(layout=cfg, k=lay.k, ndev=lay.ndev, kernel=osc.kernel_name, kernel_class=osc.kernel_class,
               algorithmic_bytes_per_step_per_instance=alg_bytes(lay))
    tgt0 = None
    for s in range(a.slots):
        q, qd = model.random_state(rng, B)
        idx = np.arange(3 + s, B, 10)                      # every 10th robot with stretched / folded arms (bench.py's workload)
        q[idx, 1:7] = (np.pi / 2) * rng.integers(-2, 3, size=(len(idx), 6))
        q[idx, 13:19] = (np.pi / 2) * rng.integers(-2, 3, size=(len(idx), 6))
        osc.upload_q(q, qd, slot=s)
        osc.frontend(slot=s)
        ee = osc.download_records(s, keys=("ee_pose",))["ee_pose"].astype(np.float64)
        tgt = synth.targets_near(ee, rng)
        osc.set_targets(tgt.astype(dt), arr.get("tgt_vel"), slot=s)
        if s == 0:
            tgt0 = tgt.astype(dt)
    osc.step_resident(max(64, a.steps // 2))
    _, ms = osc.step_resident(a.steps)
    out["physical_tree"] = dict(value=B / ms * 1e3, ms_per_step=ms, hbm_frac=alg_bytes(lay) * B / (ms * 1e-3) / 8e12,
                                tree=all(osc.slot_structure(s) for s in range(a.slots)), giveups=int(osc.giveup_counts().sum()))
    u0, f0 = osc.step(slot=0, return_flags=True)
    rec = osc.download_records(0)
    if a.oracle:
        out["physical_tree"]["max_rel_err_vs_oracle"], out["physical_tree"]["n_over_1e-5"] = oracle_err(lay, gains, rec, tgt0, None, arr.get("tgt_vel"), u0)
    out["flags"] = dict(eigen=float(((f0 & 4) != 0).mean()), truncated=float(((f0 & 8) != 0).mean()), pinv=float(((f0 & 2) != 0).mean()))
    # the fused path from joint coordinates on the same states
    if "fused" in osc.from_q_name:
        osc.step_resident_from_q(max(64, a.steps // 2))
        _, msq = osc.step_resident_from_q(a.steps)
        uq = osc.step_q(slot=0)
        d = np.abs(uq.astype(np.float64) - u0).max(axis=1) / np.maximum(np.abs(u0).max(axis=1), 1e-300)
        out["from_q"] = dict(value=B / msq * 1e3, ms_per_step=msq, max_rel_diff_vs_record_path=float(d.max()))
    osc.close()
    # synthetic dense records (no tree zeros: the dense recursion)
    osc = BatchedOSC(lay, B, dtype=dt, n_slots=a.slots)
    osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
    for s in range(a.slots):
        osc.upload(arr["M"], arr["J"], arr["dq"], arr["bias"], arr["ee_pose"], arr.get("wrench"), slot=s)
        osc.set_targets(arr["tgt_pose"], arr.get("tgt_vel"), slot=s)
    osc.step_resident(max(64, a.steps // 2))
    _, ms2 = osc.step_resident(a.steps)
    u1 = osc.step(slot=0)
    out["synthetic_dense"] = dict(value=B / ms2 * 1e3, ms_per_step=ms2, hbm_frac=alg_bytes(lay) * B / (ms2 * 1e-3) / 8e12)
    if a.oracle:
        out["synthetic_dense"]["max_rel_err_vs_oracle"], out["synthetic_dense"]["n_over_1e-5"] = oracle_err(
            lay, gains, {k: arr[k] for k in ("M", "J", "dq", "bias", "ee_pose")}, arr["tgt_pose"], arr.get("wrench"), arr.get("tgt_vel"), u1)
    osc.close()
    return out


res = []
for cfg in names:
    for fp in ([False, True] if cfg in EXACT else [False]):
        r = run(cfg, fp)
        r["forced_pad"] = fp
        res.append(r)
        fq = r.get("from_q", {}).get("value", float("nan"))
        print(f"{cfg:16s} k={r['k']:2d} ndev={r['ndev']} {r['kernel']:44s} physical+tree {r['physical_tree']['value'] / 1e6:7.1f} M/s "
              f"({r['physical_tree']['hbm_frac']:.2f} of HBM)  synthetic dense {r['synthetic_dense']['value'] / 1e6:7.1f} M/s  from_q {fq / 1e6:7.1f} M/s  "
              f"err {r['physical_tree'].get('max_rel_err_vs_oracle')} / {r['synthetic_dense'].get('max_rel_err_vs_oracle')}  giveups {r['physical_tree']['giveups']}", flush=True)
if a.out:
    with open(a.out, "w") as f:
        json.dump(dict(batch=B, steps=a.steps, dtype=a.dtype, results=res), f, indent=1)
