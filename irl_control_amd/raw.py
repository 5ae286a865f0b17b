"""Raw simulator state for the GPU-side assembly (`BatchedOSC.upload_raw` -> irlosc_upload_raw).

`Robot.get_all_states()` / `Device.get_state()` (reference robot.py:44-72,125-136, device.py:115-170) pick the
robot's rows and columns out of MuJoCo's arrays one robot at a time in Python.  For a fleet the same picking is done
by a kernel; this module only gathers the RAW arrays of many simulators into batch-major tensors and describes, once,
which entries the kernel has to pick (`raw_desc`).
"""
from typing import Dict, List, Sequence

import numpy as np

from . import _lib
from .backend import full_mass_matrix
from .device import _FT_TABLE
from .robot import Robot


def raw_desc(robot: Robot, names: Sequence[str], n_sensor: int) -> _lib.RawDesc:
    """Index tables for `names` (target devices, targets order) of `robot`."""
    d = _lib.RawDesc()
    d.nv = int(robot.num_scene_joints)
    d.n_sensor = int(n_sensor)
    ids = [int(j) for j in robot.joint_ids_all]
    n = len(ids)
    for p in range(32):
        d.joint_ids[p] = ids[p] if p < n else 0
        d.dq_src[p] = -1
    # robot.py:60-65: dq[dev.joint_ids_all] = qvel[dev.joint_ids_all] - raw joint ids used as POSITIONS in dq
    for dev in robot.sub_devices:
        for j in dev.get_all_joint_ids():
            if not 0 <= int(j) < n:
                raise IndexError("a device joint id does not fit the robot's joint vector (robot.py:63)")
            d.dq_src[int(j)] = int(j)
    for i in range(4):
        d.ft_force0[i] = d.ft_torque0[i] = -1
    for i, nm in enumerate(names):
        ent = _FT_TABLE.get(nm)
        if ent is not None:
            d.ft_force0[i], d.ft_torque0[i] = ent[1].start, ent[2].start
    return d


def collect_raw(sims: List, robots: List[Robot], names: Sequence[str], dtype=np.float64) -> Dict[str, np.ndarray]:
    """Batch-major raw arrays of len(sims) simulators (robots[i] reads sims[i]); layouts as irlosc_upload_raw."""
    B, nd = len(sims), len(names)
    nv = int(robots[0].num_scene_joints)
    ns = int(np.size(sims[0].data.sensordata))
    out = dict(qM=np.empty((B, nv, nv), dtype), qvel=np.empty((B, nv), dtype), qfrc_bias=np.empty((B, nv), dtype),
               jacp=np.empty((B, nd, 3, nv), dtype), jacr=np.empty((B, nd, 3, nv), dtype),
               ee_xpos=np.empty((B, nd, 3), dtype), ee_xquat=np.empty((B, nd, 4), dtype),
               site_xmat=np.zeros((B, nd, 9), dtype), sensordata=np.empty((B, ns), dtype))
    mvec = np.zeros(nv * nv)
    for b, (sim, rob) in enumerate(zip(sims, robots)):
        full_mass_matrix(sim, mvec)
        out["qM"][b] = mvec.reshape(nv, nv)
        out["qvel"][b] = sim.data.qvel
        out["qfrc_bias"][b] = sim.data.qfrc_bias
        out["sensordata"][b] = sim.data.sensordata
        for i, nm in enumerate(names):
            ee = rob.get_device(nm).EE
            out["jacp"][b, i] = np.asarray(sim.data.get_body_jacp(ee)).reshape(3, nv)
            out["jacr"][b, i] = np.asarray(sim.data.get_body_jacr(ee)).reshape(3, nv)
            out["ee_xpos"][b, i] = sim.data.get_body_xpos(ee)
            out["ee_xquat"][b, i] = sim.data.get_body_xquat(ee)
            ent = _FT_TABLE.get(nm)
            if ent is not None:
                out["site_xmat"][b, i] = np.asarray(sim.data.get_site_xmat(ent[0])).reshape(9)
    return out


def qm_layout(dof_parentid: Sequence[int], dof_Madr: Sequence[int] = None) -> _lib.QmLayout:
    """`struct irlosc_qm_layout` for `BatchedOSC.upload_raw(..., qm_layout=...)`: MuJoCo's sparse form of the joint-space inertia
    (mjData.qM with mjModel.dof_Madr / dof_parentid / nM; the reference expands it on the host for every robot and tick, robot.py:68-72).
    With the official bindings: qm_layout(m.dof_parentid, m.dof_Madr).  Without dof_Madr the runs are laid out back to back in dof
    order, which is what MuJoCo does."""
    par = [int(p) for p in dof_parentid]
    nv = len(par)
    if nv > _lib.MAX_NV:
        raise ValueError(f"nv = {nv} exceeds {_lib.MAX_NV}")
    q = _lib.QmLayout()
    adr = 0
    for i in range(nv):
        q.dof_parentid[i] = par[i]
        q.dof_Madr[i] = int(dof_Madr[i]) if dof_Madr is not None else adr
        j, ln = i, 0
        while j >= 0:
            ln += 1
            j = par[j]
        adr = q.dof_Madr[i] + ln
        q.nM = max(q.nM, adr)
    return q


def pack_qM(M: np.ndarray, layout: "_lib.QmLayout") -> np.ndarray:
    """Dense [B, nv, nv] -> MuJoCo's sparse [B, nM] (the inverse of mj_fullM; entries outside the tree's pattern are dropped -- a
    simulator holds the sparse form to begin with, this is for tests and the bench)."""
    M = np.asarray(M)
    B, nv = M.shape[0], M.shape[1]
    out = np.zeros((B, int(layout.nM)), dtype=M.dtype)
    for i in range(nv):
        adr, j = int(layout.dof_Madr[i]), i
        while j >= 0:
            out[:, adr] = M[:, i, j]
            adr += 1
            j = int(layout.dof_parentid[j])
    return out
