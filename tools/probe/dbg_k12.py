import sys, numpy as np
sys.path.insert(0, '.')
from irl_control_amd import BatchedOSC, synth, _lib
for cfg in ['k12_admit', 'k13']:
    lay, gains, g = synth.make_batch(cfg, 1024, seed=99)
    g32 = {k: v.astype(np.float32) for k, v in g.items()}
    res = {}
    for kern in (_lib.KERNEL_GENERIC, _lib.KERNEL_AUTO):
        osc = BatchedOSC(lay, 1024, dtype=np.float32, kernel=kern)
        osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
        u, fl = osc.generate_batched(g32["M"], g32["J"], g32["dq"], g32["bias"], g32["ee_pose"], g32["tgt_pose"], g32.get("tgt_vel"), g32.get("wrench"), return_flags=True)
        res[kern] = (u, fl); print(cfg, osc.kernel_name, 'nan rows', np.isnan(u).any(axis=1).sum(), 'flag hist', {int(f): int((fl == f).sum()) for f in np.unique(fl)})
        osc.close()
    u0, f0 = res[1]; u1, f1 = res[0]
    bad = np.where(np.isnan(u1).any(axis=1))[0]
    print(' bad idx', bad[:10], 'flags', f1[bad[:10]], 'generic flags', f0[bad[:10]])
    if len(bad): print(u1[bad[0]], u0[bad[0]])
    d = np.abs(u1 - u0).max(axis=1) / np.abs(u0).max(axis=1)
    print(' group vs generic rel diff: median %.2e p99 %.2e max %.2e' % (np.nanmedian(d), np.nanquantile(d, .99), np.nanmax(d)))
