"""``BatchedOSC``: many independent robot instances per control tick through libirlosc.

This is the batch-of-B form of ``OSC.generate`` (/root/reference/irl_control/osc.py:120-210): the
same inputs the reference assembles per tick (M, stacked J, dq, bias, EE poses, targets, wrench),
with a leading batch axis, in; joint torques ``u_all`` [B, n] out.  All arithmetic happens in the
HIP kernels; this class only owns the context and checks shapes.
"""
import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from .layout import OSCLayout, pack_gains


class BatchedOSC:
    def __init__(self, layout: OSCLayout, max_batch: int, dtype=np.float64, hip_device: int = 0,
                 n_slots: int = 1, kernel: int = _lib.KERNEL_AUTO):
        self.lib = _lib.load()
        self.layout = layout
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        self.max_batch, self.n_slots = int(max_batch), int(n_slots)
        code = _lib.F64 if self.dtype == np.float64 else _lib.F32
        cfg = layout.to_cfg(code, self.max_batch, hip_device, self.n_slots, kernel)
        h = C.c_void_p()
        rc = self.lib.irlosc_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise _lib.IrloscError(f"irlosc_create failed ({rc}): "
                                   f"{self.lib.irlosc_last_error(None).decode()}")
        self._h = h
        self._B = [0] * self.n_slots
        if kernel == _lib.KERNEL_AUTO and self.max_batch >= 1024 and self.kernel_class == _lib.CLASS_GENERIC:
            import warnings
            warnings.warn(f"layout (n={layout.n}, k={layout.k}, ndev={layout.ndev}) has no throughput kernel: "
                          f"{self.kernel_name} (one wavefront per instance, ~25x slower than osc_row16 at this batch size); "
                          "the row16 kernels take every layout of an n=25 robot (k <= 16, ndev <= 4)", RuntimeWarning, stacklevel=2)

    # -- plumbing ---------------------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise _lib.IrloscError(f"libirlosc error {rc}: {self.lib.irlosc_last_error(self._h).decode()}")

    def _arr(self, a, shape, name):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.shape != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {a.shape}")
        return a

    @staticmethod
    def _check_symmetric(M):
        """The throughput kernels read rows of M as columns (include/irlosc.h): refuse an asymmetric inertia matrix."""
        asym = np.abs(M - np.swapaxes(M, 1, 2)).max(axis=(1, 2))
        scale = np.abs(M).max(axis=(1, 2))
        bad = np.nonzero(asym > 1e-6 * np.maximum(scale, 1e-300))[0]
        if len(bad):
            raise ValueError(f"M of instance {int(bad[0])} is not symmetric (max |M - M^T| = {asym[bad[0]]:.3g})")

    @property
    def kernel_name(self) -> str:
        return self.lib.irlosc_kernel_name(self._h).decode()

    @property
    def kernel_class(self) -> int:
        """_lib.CLASS_GENERIC / CLASS_ROW16 (an instantiation for exactly this layout) / CLASS_ROW16_PADDED (row16 kernel of the
        next tier KMAX >= k, k and ndev at run time) / CLASS_GROUP: irlosc_kernel_class."""
        return int(self.lib.irlosc_kernel_class(self._h))

    @property
    def frontend_name(self) -> str:
        """Kernel `frontend()` launches: the lane-per-robot one when the model has the compiled Dual-UR5 shape, else the generic one."""
        return self.lib.irlosc_frontend_name(self._h).decode()

    def close(self):
        if getattr(self, "_h", None):
            self.lib.irlosc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- API ---------------------------------------------------------------------------------------
    def set_gains(self, kp, kv, ko, k, d, max_vel, null_kv=0.0):
        g, nk, nb = pack_gains(self.layout, kp, kv, ko, k, d, max_vel, null_kv)
        if nb not in (1, self.max_batch):
            raise ValueError(f"per-instance gains need a leading axis of max_batch={self.max_batch}")
        last = getattr(self, "_gains_sent", None)      # a per-tick caller re-sends the same gains: skip the copy then
        if last is not None and last[2] == nb and np.array_equal(last[0], g) and np.array_equal(last[1], nk):
            return
        self._chk(self.lib.irlosc_set_gains(self._h, _lib.ptr(g), _lib.ptr(nk), nb))
        self._gains_sent = (g.copy(), nk.copy(), nb)

    def upload(self, M, J, dq, bias, ee_pose, wrench=None, slot: int = 0, check_symmetric: bool = False):
        """Records of B robots into resident slot `slot`.  On the throughput kernels the library itself refuses an asymmetric M
        (device-side probe, IrloscError); check_symmetric=True additionally checks on the host before anything is copied."""
        L = self.layout
        B = int(np.shape(M)[0])
        M = self._arr(M, (B, L.n, L.n), "M")
        if check_symmetric:
            self._check_symmetric(M)
        J = self._arr(J, (B, L.k, L.n), "J")
        dq = self._arr(dq, (B, L.n), "dq")
        bias = self._arr(bias, (B, L.n), "bias")
        ee = self._arr(ee_pose, (B, L.ndev, 7), "ee_pose")
        wr = self._arr(wrench, (B, L.ndev, 6), "wrench")
        self._chk(self.lib.irlosc_upload(self._h, slot, B, _lib.ptr(M), _lib.ptr(J), _lib.ptr(dq),
                                         _lib.ptr(bias), _lib.ptr(ee), _lib.ptr(wr)))
        self._B[slot] = B

    def upload_raw(self, desc, qM, qvel, qfrc_bias, jacp, jacr, ee_xpos, ee_xquat, site_xmat=None, sensordata=None,
                   slot: int = 0, qm_layout=None):
        """Raw simulator arrays in, state assembly on the GPU (see raw.py / irlosc_upload_raw).  `desc` from
        raw.raw_desc(); arrays batch-major: qM[B,nv,nv], qvel[B,nv], qfrc_bias[B,nv], jacp/jacr[B,ndev,3,nv],
        ee_xpos[B,ndev,3], ee_xquat[B,ndev,4], site_xmat[B,ndev,9], sensordata[B,n_sensor].
        With `qm_layout` (raw.qm_layout(dof_parentid[, dof_Madr])) qM is mjData.qM as MuJoCo holds it, [B, nM]: mj_fullM's expansion
        (robot.py:68-72) then runs on the GPU (irlosc_upload_raw_sparse)."""
        L = self.layout
        B, nv, ns = int(np.shape(qM)[0]), int(desc.nv), int(desc.n_sensor)
        if qm_layout is not None:
            a = [self._arr(qM, (B, int(qm_layout.nM)), "qM (sparse)"), self._arr(qvel, (B, nv), "qvel"), self._arr(qfrc_bias, (B, nv), "qfrc_bias"),
                 self._arr(jacp, (B, L.ndev, 3, nv), "jacp"), self._arr(jacr, (B, L.ndev, 3, nv), "jacr"),
                 self._arr(ee_xpos, (B, L.ndev, 3), "ee_xpos"), self._arr(ee_xquat, (B, L.ndev, 4), "ee_xquat"),
                 self._arr(site_xmat, (B, L.ndev, 9), "site_xmat"), self._arr(sensordata, (B, ns), "sensordata")]
            self._chk(self.lib.irlosc_upload_raw_sparse(self._h, slot, B, C.byref(desc), C.byref(qm_layout), *[_lib.ptr(x) for x in a]))
            self._B[slot] = B
            return
        a = [self._arr(qM, (B, nv, nv), "qM"), self._arr(qvel, (B, nv), "qvel"), self._arr(qfrc_bias, (B, nv), "qfrc_bias"),
             self._arr(jacp, (B, L.ndev, 3, nv), "jacp"), self._arr(jacr, (B, L.ndev, 3, nv), "jacr"),
             self._arr(ee_xpos, (B, L.ndev, 3), "ee_xpos"), self._arr(ee_xquat, (B, L.ndev, 4), "ee_xquat"),
             self._arr(site_xmat, (B, L.ndev, 9), "site_xmat"), self._arr(sensordata, (B, ns), "sensordata")]
        self._chk(self.lib.irlosc_upload_raw(self._h, slot, B, C.byref(desc), *[_lib.ptr(x) for x in a]))
        self._B[slot] = B

    def set_targets(self, tgt_pose, tgt_vel=None, slot: int = 0):
        L = self.layout
        B = int(np.shape(tgt_pose)[0])
        if self._B[slot] and B != self._B[slot]:
            raise ValueError(f"slot {slot} holds {self._B[slot]} instances, targets given for {B}")
        tp = self._arr(tgt_pose, (B, L.ndev, 7), "tgt_pose")
        tv = self._arr(tgt_vel, (B, L.ndev, 6), "tgt_vel")
        self._chk(self.lib.irlosc_set_targets(self._h, slot, B, _lib.ptr(tp), _lib.ptr(tv)))

    def step(self, slot: int = 0, return_flags: bool = False):
        B = self._B[slot]
        u = np.empty((B, self.layout.n), dtype=self.dtype)
        fl = np.empty(B, dtype=np.uint32)
        self._chk(self.lib.irlosc_step(self._h, slot, B, _lib.ptr(u), _lib.ptr(fl)))
        return (u, fl) if return_flags else u

    def step_resident(self, iters: int, first_slot: int = 0, B: Optional[int] = None):
        """-> (ms_total, ms_kernel_avg): `iters` launches on resident data, HIP-event timed."""
        B = self._B[first_slot] if B is None else B
        t, a = C.c_float(), C.c_float()
        self._chk(self.lib.irlosc_step_resident(self._h, first_slot, B, iters, C.byref(t), C.byref(a)))
        return t.value, a.value

    def time_dominant_kernel(self, iters: int, slot: int = 0, B: Optional[int] = None) -> float:
        """Mean duration (ms) of the dominant kernel alone over `iters` launches (roofline support)."""
        B = self._B[slot] if B is None else B
        a = C.c_float()
        self._chk(self.lib.irlosc_time_dominant_kernel(self._h, slot, B, iters, C.byref(a)))
        return a.value

    def time_trains(self, ntrains: int, first_slot: int = 0, B: Optional[int] = None, from_q: bool = False) -> np.ndarray:
        """Untraced timing of `ntrains` consecutive trains (irlosc_time_trains): -> [ntrains, 4] = (HIP event pair of the train in
        ms, start of its first wave, end of its last wave in microseconds of the kernels' own 100 MHz clock since train 0, shader
        clock in MHz seen by a sample wave of the train)."""
        B = self._B[first_slot] if B is None else B
        out = np.zeros((int(ntrains), 4), dtype=np.float64)
        self._chk(self.lib.irlosc_time_trains(self._h, first_slot, B, int(ntrains), 1 if from_q else 0, _lib.ptr(out)))
        return out

    def giveup_counts(self) -> np.ndarray:
        """Instances the steps of the most recent train (a plain step: entry 0) handed to the generic kernel's Jacobi solve
        (irlosc_giveup_counts): same results, but a serial tail -- watch it when task sets can lose more than three directions."""
        out = np.zeros(8, dtype=np.int32)
        self._chk(self.lib.irlosc_giveup_counts(self._h, _lib.ptr(out)))
        return out

    @property
    def steps_per_launch(self) -> int:
        """Steps chained into one launch by step_resident / time_dominant_kernel (1 on the generic path)."""
        n = int(self.lib.irlosc_steps_per_launch(self._h))
        if n < 1:
            raise _lib.IrloscError(f"libirlosc error {n}")
        return n

    def download(self, B: Optional[int] = None):
        B = self._B[0] if B is None else B
        u = np.empty((B, self.layout.n), dtype=self.dtype)
        fl = np.empty(B, dtype=np.uint32)
        self._chk(self.lib.irlosc_download(self._h, B, _lib.ptr(u), _lib.ptr(fl)))
        return u, fl

    # -- rigid-body front end: joint coordinates in, records assembled on the GPU ------------------------------------
    def set_model(self, model, ee_bodies=None):
        """`model`: rigid_body.RigidBodyModel; `ee_bodies`: EE body name per target device (targets order), default the
        Dual-UR5 map."""
        from .rigid_body import DUAL_UR5_EE
        names = ee_bodies if ee_bodies is not None else [DUAL_UR5_EE[d] for d in self.layout.dev_names]
        st = model.to_struct(list(names))
        self._chk(self.lib.irlosc_set_model(self._h, C.byref(st)))
        self._model = model

    def upload_q(self, qpos, qvel, slot: int = 0):
        B = int(np.shape(qpos)[0])
        qp = np.ascontiguousarray(qpos, dtype=np.float64)
        qv = np.ascontiguousarray(qvel, dtype=np.float64)
        if qp.shape != (B, self.layout.n) or qv.shape != (B, self.layout.n):
            raise ValueError(f"qpos / qvel: expected shape {(B, self.layout.n)}")
        self._chk(self.lib.irlosc_upload_q(self._h, slot, B, _lib.ptr(qp), _lib.ptr(qv)))
        self._B[slot] = B

    def frontend(self, slot: int = 0):
        """(qpos, qvel) of the slot -> its M, J, dq, bias, ee_pose records (on the GPU)."""
        self._chk(self.lib.irlosc_frontend(self._h, slot, self._B[slot]))

    def download_records(self, slot: int = 0, keys=("M", "J", "dq", "bias", "ee_pose")):
        """-> dict of the slot's records as they sit in HBM (uploaded, or assembled by the front end); `keys` picks which
        ones cross PCIe (a per-tick caller that only needs the EE poses asks for ("ee_pose",): 168 B instead of 8.5 KB per robot)."""
        L, B = self.layout, self._B[slot]
        shapes = dict(M=(B, L.n, L.n), J=(B, L.k, L.n), dq=(B, L.n), bias=(B, L.n), ee_pose=(B, L.ndev, 7))
        out = {k: np.empty(shapes[k], self.dtype) for k in keys}
        self._chk(self.lib.irlosc_download_records(self._h, slot, B, *[_lib.ptr(out.get(k)) for k in ("M", "J", "dq", "bias", "ee_pose")]))
        return out

    def slot_structure(self, slot: int = 0) -> bool:
        """True when the records in `slot` carry the zero pattern of the Dual-UR5 tree and the fp64 row16 kernel therefore
        factors M in the tree-structured form (irlosc_slot_structure, include/irlosc.h)."""
        return bool(self.lib.irlosc_slot_structure(self._h, slot))

    def probe_structure(self, slot: int = 0, B: Optional[int] = None) -> bool:
        """Look at the records already in `slot` (irlosc_probe_structure): what assemble_device on a caller's stream cannot
        do for itself.  -> the slot's verdict, as slot_structure() reports it from then on."""
        rc = self.lib.irlosc_probe_structure(self._h, slot, self._B[slot] if B is None else B)
        if rc < 0:
            self._chk(rc)
        return bool(rc)

    @property
    def from_q_name(self) -> str:
        """What step_from_q / step_resident_from_q launch: the fused pair (compact exchange buffer, no dense M / J) when the
        model has the compiled Dual-UR5 shape and the context runs the fp64 row16 kernel, else front end + step."""
        return self.lib.irlosc_from_q_name(self._h).decode()

    def step_q(self, slot: int = 0, return_flags: bool = False):
        """One step from the slot's resident (qpos, qvel) and targets (irlosc_step_from_q)."""
        B = self._B[slot]
        u = np.empty((B, self.layout.n), dtype=self.dtype)
        fl = np.empty(B, dtype=np.uint32)
        self._chk(self.lib.irlosc_step_from_q(self._h, slot, B, _lib.ptr(u), _lib.ptr(fl)))
        return (u, fl) if return_flags else u

    def step_from_q(self, qpos, qvel, tgt_pose, tgt_vel=None, return_flags: bool = False):
        """One tick from joint coordinates: upload (qpos, qvel) and the targets, one step from them, download."""
        self.upload_q(qpos, qvel)
        self.set_targets(tgt_pose, tgt_vel)
        return self.step_q(return_flags=return_flags)

    def step_resident_from_q(self, iters: int, first_slot: int = 0, B: Optional[int] = None):
        """-> (ms_total, ms_per_step): `iters` x (front end + step) on resident joint coordinates, HIP-event timed."""
        B = self._B[first_slot] if B is None else B
        t, a = C.c_float(), C.c_float()
        self._chk(self.lib.irlosc_step_resident_from_q(self._h, first_slot, B, iters, C.byref(t), C.byref(a)))
        return t.value, a.value

    def sync(self):
        self._chk(self.lib.irlosc_sync(self._h))

    def device_sync(self):
        """hipDeviceSynchronize on this context's GPU (the bench bracket)."""
        self._chk(self.lib.irlosc_device_sync(self._h))

    def tick(self, M, J, dq, bias, ee_pose, tgt_pose, tgt_vel=None, wrench=None, return_flags: bool = False,
             check_symmetric: bool = False):
        """One control tick for B instances in ONE library call (irlosc_tick): one host-to-device copy, the step, one
        copy back, one synchronisation.  Does not touch the resident slots."""
        L = self.layout
        B = int(np.shape(M)[0])
        M = self._arr(M, (B, L.n, L.n), "M")
        if check_symmetric:
            self._check_symmetric(M)
        J = self._arr(J, (B, L.k, L.n), "J")
        dq = self._arr(dq, (B, L.n), "dq")
        bias = self._arr(bias, (B, L.n), "bias")
        ee = self._arr(ee_pose, (B, L.ndev, 7), "ee_pose")
        wr = self._arr(wrench, (B, L.ndev, 6), "wrench")
        tp = self._arr(tgt_pose, (B, L.ndev, 7), "tgt_pose")
        tv = self._arr(tgt_vel, (B, L.ndev, 6), "tgt_vel")
        u = np.empty((B, L.n), dtype=self.dtype)
        fl = np.empty(B, dtype=np.uint32)
        self._chk(self.lib.irlosc_tick(self._h, B, _lib.ptr(M), _lib.ptr(J), _lib.ptr(dq), _lib.ptr(bias), _lib.ptr(ee),
                                       _lib.ptr(wr), _lib.ptr(tp), _lib.ptr(tv), _lib.ptr(u), _lib.ptr(fl)))
        return (u, fl) if return_flags else u

    def generate_batched(self, M, J, dq, bias, ee_pose, tgt_pose, tgt_vel=None, wrench=None,
                         return_flags: bool = False):
        """One tick for B instances: upload, step, download."""
        self.upload(M, J, dq, bias, ee_pose, wrench)
        self.set_targets(tgt_pose, tgt_vel)
        return self.step(return_flags=return_flags)
