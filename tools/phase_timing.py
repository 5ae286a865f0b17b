#!/usr/bin/env python3
"""Mean cycles per kernel phase (IRLOSC_PHASE_TIMING=1 debug aid of libirlosc).  usage: phase_timing.py [f32|f64] [layout] [B] [phys|fromq]
"phys": records of physical robot states (front end), which qualify for the tree-structured form (IRLOSC_TREE=0: dense);
"fromq": the fused path from joint coordinates (the OSC kernel's first phase then includes the fill of its LDS tile)."""
import os
import sys
os.environ["IRLOSC_PHASE_TIMING"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from irl_control_amd import BatchedOSC, synth

dt = np.float64 if (len(sys.argv) < 2 or sys.argv[1] == "f64") else np.float32
cfg = sys.argv[2] if len(sys.argv) > 2 else "k13"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
lay, gains, arr = synth.make_batch(cfg, B, seed=7, dtype=dt)
osc = BatchedOSC(lay, B, dtype=dt)
osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
mode = sys.argv[4] if len(sys.argv) > 4 else ""
if mode in ("phys", "fromq"):
    from irl_control_amd.rigid_body import RigidBodyModel
    model = RigidBodyModel.load("dual_ur5")
    rng = np.random.default_rng(5)
    osc.set_model(model)
    osc.upload_q(*model.random_state(rng, B))
    osc.frontend()
    ee = osc.download_records(0, keys=("ee_pose",))["ee_pose"].astype(np.float64)
    osc.set_targets(synth.targets_near(ee, rng).astype(dt))
    print("tree form:", osc.slot_structure(0))
else:
    osc.upload(arr["M"], arr["J"], arr["dq"], arr["bias"], arr["ee_pose"], arr.get("wrench"))
    osc.set_targets(arr["tgt_pose"], arr.get("tgt_vel"))
for _ in range(3):
    if mode == "fromq":
        osc.step_resident_from_q(48)
    else:
        osc.step_resident(50)
    osc.download(B)       # prints the phase table to stderr
print(osc.from_q_name if mode == "fromq" else osc.kernel_name)
