import time, numpy as np, sys
sys.path.insert(0, '.')
from irl_control_amd import BatchedOSC, synth
for B in (1, 16, 512, 4096):
    lay, gains, g = synth.make_batch("k13", max(B, 16), seed=1, dtype=np.float32)
    g = {k: (v[:B] if isinstance(v, np.ndarray) else v) for k, v in g.items()}
    osc = BatchedOSC(lay, B, dtype=np.float32)
    osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
    osc.upload(g["M"], g["J"], g["dq"], g["bias"], g["ee_pose"], g.get("wrench")); osc.set_targets(g["tgt_pose"])
    for _ in range(5): osc.step()
    t0 = time.perf_counter()
    for _ in range(200): osc.step()
    dt = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(50):
        osc.upload(g["M"], g["J"], g["dq"], g["bias"], g["ee_pose"], g.get("wrench")); osc.set_targets(g["tgt_pose"]); osc.step()
    dt2 = (time.perf_counter() - t0) / 50
    print(f"B={B}: step (launch+download+sync) {dt*1e6:.0f} us; upload+targets+step {dt2*1e6:.0f} us  kernel={osc.kernel_name}")
    osc.close()
