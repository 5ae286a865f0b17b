#!/usr/bin/env python3
"""Wide check of the FUSED path from joint coordinates (lane-per-robot walk -> compact exchange buffer -> row16 kernel
gathering its operands) against the path through dense records (record front end + row16 kernel on the records): the same
arithmetic on the same numbers, so torques must agree to rounding and flags exactly -- on 65 536 robots per seed and layout:
random joint angles over many turns, velocities up to +-20 rad/s, a quarter of the robots with every arm angle a multiple
of pi / 2 (stretched / folded arms: rank-deficient task Jacobians, the eigen stage and the give-up hand-over), random
targets; plus the float64 torques against the GENERIC kernel (Jacobi) on the records as an independent implementation.
    python tools/fused_sweep.py [--seeds 3] [--batch 65536]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irl_control_amd import BatchedOSC, _lib, synth                # noqa: E402
from irl_control_amd.rigid_body import RigidBodyModel              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=3)
ap.add_argument("--batch", type=int, default=65536)
a = ap.parse_args()
B = a.batch
model = RigidBodyModel.load("dual_ur5")
tot = dict(robots=0, flag_mismatch=0, over_1e9=0, over_1e5_vs_generic=0, giveups=0, eigen=0)
worst, worst_g = 0.0, 0.0
for layout in ("k13", "k12_admit", "k7"):
    lay = synth.make_layout(layout)
    for sd in range(a.seeds):
        rng = np.random.default_rng(4242 + sd)
        _, gains, g = synth.make_batch(layout, B, seed=900 + sd)
        q = rng.uniform(-np.pi, np.pi, (B, lay.n)) * rng.choice([1.0, 1.0, 7.0, 300.0], size=(B, 1))
        qd = rng.normal(0.0, 1.0, (B, lay.n)) * rng.choice([0.0, 1.0, 5.0, 20.0], size=(B, 1))
        sing = np.arange(sd % 4, B, 4)
        q[sing, 1:7] = (np.pi / 2) * rng.integers(-2, 3, size=(len(sing), 6))
        q[sing, 13:19] = (np.pi / 2) * rng.integers(-2, 3, size=(len(sing), 6))
        osc = BatchedOSC(lay, B, dtype=np.float64, kernel=_lib.KERNEL_ROW16)
        osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
        osc.set_model(model)
        assert "fused" in osc.from_q_name
        osc.upload_q(q, qd)
        osc.set_targets(g["tgt_pose"], g.get("tgt_vel"))
        if "wrench" in g:       # admittance: the wrench is not a function of (q, qd); it comes with an upload of records
            osc.frontend()
            rec = osc.download_records()
            osc.upload(rec["M"], rec["J"], rec["dq"], rec["bias"], rec["ee_pose"], g["wrench"])
        u_f, fl_f = osc.step_q(return_flags=True)
        osc.frontend()
        u_d, fl_d = osc.step(return_flags=True)
        rec = osc.download_records()
        osc.close()
        gen = BatchedOSC(lay, B, dtype=np.float64, kernel=_lib.KERNEL_GENERIC)
        gen.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
        u_g, fl_g = gen.generate_batched(rec["M"], rec["J"], rec["dq"], rec["bias"], rec["ee_pose"], g["tgt_pose"], g.get("tgt_vel"),
                                         g.get("wrench"), return_flags=True)
        gen.close()
        d = np.abs(u_f - u_d).max(axis=1) / np.abs(u_d).max(axis=1)
        dg = np.abs(u_f - u_g).max(axis=1) / np.abs(u_g).max(axis=1)
        same_cut = ((fl_f ^ fl_g) & _lib.FLAG_TRUNCATED) == 0            # the generic kernel's exact spectrum decides the same cut
        tot["robots"] += B
        tot["flag_mismatch"] += int((fl_f != fl_d).sum())
        tot["over_1e9"] += int((d > 1e-9).sum())
        tot["over_1e5_vs_generic"] += int(((dg > 1e-5) & same_cut).sum())
        tot["eigen"] += int(((fl_f & _lib.FLAG_EIGEN_PATH) != 0).sum())
        worst, worst_g = max(worst, float(d.max())), max(worst_g, float(dg[same_cut].max()))
        print(f"{layout} seed {sd}: fused vs dense-record path max {d.max():.2e}, flags differing {int((fl_f != fl_d).sum())}; "
              f"vs generic kernel (same cut decision: {int(same_cut.sum())}) max {dg[same_cut].max():.2e}, over 1e-5: "
              f"{int(((dg > 1e-5) & same_cut).sum())}; eigen stage {((fl_f & 4) != 0).mean():.3f}, truncated {((fl_f & 8) != 0).mean():.3f}, "
              f"non-finite {int(((fl_f & 64) != 0).sum())}", flush=True)
print("TOTAL", tot, f"worst fused-vs-dense {worst:.2e}, worst vs generic {worst_g:.2e}")
