import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The shared library is a build product (git-ignored): compile it when it is missing, so that a fresh checkout
    can run the suite without a separate build step.  hipcc cross-compiles gfx950 without a GPU (~1.5 min).  A box
    without hipcc (the GPU box gets the prebuilt .so with the snapshot) simply uses what is there."""
    lib = os.path.join(ROOT, "irl_control_amd", "libirlosc.so")
    if not os.path.exists(lib) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        import __graft_entry__
        __graft_entry__.build()


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and not f.startswith(("e2e_", "loop_")))


def load_e2e(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files if k != "layout_json"}
    g["meta"] = json.loads(str(z["layout_json"]))
    return g


def sim_from_e2e(g, b):
    """Rebuild the FakeSim state of instance b of an end-to-end fixture."""
    from irl_control_amd.fakesim import FakeSim
    meta = g["meta"]
    sim = FakeSim(n_free_bodies=meta["n_free_bodies"])
    d = sim.data
    d.qM = g["qM"][b].copy()
    d.qvel[:] = g["qvel"][b]
    d.qfrc_bias[:] = g["qfrc_bias"][b]
    d.sensordata[:] = g["sensordata"][b]
    for i, body in enumerate(meta["ee_bodies"]):
        bid = sim.model.body_name2id(body)
        d.body_jacp[bid] = g["jacp"][b, i]
        d.body_jacr[bid] = g["jacr"][b, i]
        d.body_xpos[bid] = g["xpos"][b, i]
        d.body_xquat[bid] = g["xquat"][b, i]
    d.site_xmat["ft_frame_ur5right"] = g["xmat_right"][b].copy()
    d.site_xmat["ft_frame_ur5left"] = g["xmat_left"][b].copy()
    return sim


def app_from_e2e(g, b):
    """-> (app, robot, osc, targets) built with the build's own classes on the fixture's state."""
    import irl_control_amd as ic
    meta = g["meta"]
    sim = sim_from_e2e(g, b)
    app = ic.MujocoApp(meta["cfg_file"], None, sim=sim)
    robot = app.get_robot("DualUR5")
    cfgs = [(dn, app.get_controller_config(gn)) for dn, gn in meta["dev_gain_names"]]
    ns = app.get_controller_config("nullspace") if meta["nullspace"] else None
    osc = ic.OSC(robot, sim, cfgs, ns, use_g=meta["use_g"], admittance=meta["admittance"])
    targets = {}
    for i, dn in enumerate(meta["target_order"]):
        t = ic.Target()
        t.set_all_quat(g["tgt_xyz"][b, i], g["tgt_quat"][b, i])
        targets[dn] = t
    return app, robot, osc, targets


def live_mutation_phases(g, b):
    """e2e_live_mutations fixture: build the build's own app on instance b, then yield (phase, app, robot, osc, targets)
    with each phase's mutations applied to the LIVE devices first (ctrlr_dof_abg re-set as examples/ps_move_example.py:137-150
    does, max_vel[0] as examples/insertion_task.py:294 does) -- one OSC object for all phases."""
    import irl_control_amd as ic
    meta = g["meta"]
    g0 = dict(g, tgt_xyz=g["tgt_xyz"][:, 0], tgt_quat=g["tgt_quat"][:, 0])
    app, robot, osc, _ = app_from_e2e(g0, b)
    for p, ph in enumerate(meta["phases"]):
        for dn, mask in ph.get("abg", {}).items():
            robot.get_device(dn).ctrlr_dof_abg = list(mask)
        for dn, v in ph.get("max_vel0", {}).items():
            robot.get_device(dn).max_vel[0] = v
        targets = {}
        for i, dn in enumerate(meta["target_order"]):
            t = ic.Target()
            t.set_all_quat(g["tgt_xyz"][b, p, i], g["tgt_quat"][b, p, i])
            targets[dn] = t
        yield p, app, robot, osc, targets


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files if k != "layout_json"}
    g["layout"] = json.loads(str(z["layout_json"]))
    return g


def golden_gains(g):
    """gains dict (per-instance arrays) in the form oracle.generate_batch / BatchedOSC accept."""
    return dict(kp=g["kp"], kv=g["kv"], ko=g["ko"], k=g["kk"], d=g["dd"], max_vel=g["max_vel"],
                null_kv=g["null_kv"])


def golden_expected_u(g):
    """Scatter the reference's per-device forces back into the n-vector; NaN where no actuator."""
    lay = g["layout"]
    B = g["M"].shape[0]
    u = np.full((B, lay["n"]), np.nan)
    off = 0
    for trn in lay["actuator_trnids"]:
        u[:, trn] = g["forces_flat"][:, off:off + len(trn)]
        off += len(trn)
    return u


@pytest.fixture(scope="session")
def goldens():
    return {n: load_golden(n) for n in golden_names()}


class HipBuffers:
    """Device buffers for the tests that hand DEVICE pointers to the C ABI (irlosc_step_device / irlosc_assemble_device), straight
    from libamdhip64 through ctypes: hipMalloc / hipMemcpy / hipFree -- no framework in between, so these tests cannot silently
    vanish on a box without one."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self._ptrs = []

    def _chk(self, rc, what):
        assert rc == 0, f"{what} failed with hipError {rc}"

    def alloc(self, nbytes):
        p = self.C.c_void_p()
        self._chk(self.hip.hipMalloc(self.C.byref(p), max(int(nbytes), 16)), "hipMalloc")
        self._ptrs.append(p)
        return p

    def to_device(self, a):
        a = np.ascontiguousarray(a)
        p = self.alloc(a.nbytes)
        self._chk(self.hip.hipMemcpy(p, a.ctypes.data_as(self.C.c_void_p), a.nbytes, 1), "hipMemcpy H2D")      # hipMemcpyHostToDevice
        return p

    def to_host(self, p, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._chk(self.hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
        self._chk(self.hip.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), p, out.nbytes, 2), "hipMemcpy D2H")   # hipMemcpyDeviceToHost
        return out

    def free(self):
        for p in self._ptrs:
            self.hip.hipFree(p)
        self._ptrs = []


# ---- the oracle on EVERY instance of a large batch: forked workers over the usable host cores (the arrays are inherited, not pickled) ----
_ORACLE_JOB = {}


def _oracle_chunk(span):
    import numpy as np
    from oracle import osc_oracle
    j = _ORACLE_JOB
    lo, hi = span
    r = {k: np.ascontiguousarray(v[lo:hi], dtype=np.float64) for k, v in j["rec"].items()}
    ref = osc_oracle.generate_batch(j["lay"], j["gains"], r["M"], r["J"], r["dq"], r["bias"], r["ee_pose"], r["tgt_pose"],
                                    r.get("wrench"), r.get("tgt_vel"))
    dom, pinv, trunc, dets = np.zeros(hi - lo, bool), np.zeros(hi - lo, bool), np.zeros(hi - lo, bool), np.zeros(hi - lo)
    for b in range(hi - lo):
        Mx, Minv, Mxi, det = osc_oracle.task_inertia(r["J"][b], r["M"][b])
        sv = np.linalg.svd(Mxi, compute_uv=False)
        if abs(det) >= 1e-4:
            dom[b] = sv[-1] > 1e-12 * sv[0]
        else:
            dom[b] = not np.any(np.abs(sv / sv[0] / 1e-5 - 1.0) < 1e-2)
        pinv[b] = abs(det) < 1e-4
        dets[b] = det
        trunc[b] = abs(det) < 1e-4 and sv[-1] <= 1e-5 * sv[0]
    return lo, ref, dom, pinv, trunc, dets


def oracle_on_all(lay_dict, gains, rec):
    """-> (ref[B, n], in_parity_domain[B], pinv_branch[B], truncates[B], det[B]): oracle/osc_oracle.generate_batch + the reference's branch
    (osc.py:51-55) for every instance of `rec` (C-ABI arrays incl. tgt_pose), over the cores this process may use."""
    import multiprocessing as mp
    import os

    import numpy as np
    B = rec["M"].shape[0]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                cores = max(1, min(cores, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    nw = max(1, min(cores, 32))
    _ORACLE_JOB.update(lay=lay_dict, gains=gains, rec=rec)
    step = max(64, -(-B // (nw * 8)))
    spans = [(lo, min(B, lo + step)) for lo in range(0, B, step)]
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except ImportError:
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        if nw == 1:
            parts = [_oracle_chunk(sp) for sp in spans]
        else:
            with mp.get_context("fork").Pool(nw) as pool:
                parts = pool.map(_oracle_chunk, spans)
    _ORACLE_JOB.clear()
    n = parts[0][1].shape[1]
    ref, dom, pinv, trunc, det = np.zeros((B, n)), np.zeros(B, bool), np.zeros(B, bool), np.zeros(B, bool), np.zeros(B)
    for lo, r, d, p_, t, dt in parts:
        ref[lo:lo + len(d)], dom[lo:lo + len(d)], pinv[lo:lo + len(d)], trunc[lo:lo + len(d)], det[lo:lo + len(d)] = r, d, p_, t, dt
    return ref, dom, pinv, trunc, det
