"""``OSCLayout``: the static description of one OSC problem shape — which devices are targeted, in
which order, which task rows they control, which joints they own — i.e. everything the reference
derives from its Device/Robot objects at every tick (device.py:36,66-69; robot.py:28-32,50-55;
osc.py:136-138) and that the C ABI takes once in ``irlosc_cfg``.
"""
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

from . import _lib

GAIN_WORDS = _lib.GAIN_WORDS


@dataclass
class OSCLayout:
    n: int
    dev_names: List[str]
    ctrlr_dof: List[List[bool]]            # [ndev][6]
    joint_ids: List[List[int]]             # positions in the n-vector per device
    j_idx0: List[int]
    calc_xyz: List[bool] = field(default_factory=list)
    calc_abg: List[bool] = field(default_factory=list)
    has_max_vel: List[bool] = field(default_factory=list)
    use_g: bool = True
    admittance: bool = False
    nullspace: bool = True

    def __post_init__(self):
        nd = len(self.dev_names)
        if not self.calc_xyz:
            self.calc_xyz = [bool(np.sum(m[:3]) > 0) for m in self.ctrlr_dof]
        if not self.calc_abg:
            self.calc_abg = [bool(np.sum(m[3:]) > 0) for m in self.ctrlr_dof]
        if not self.has_max_vel:
            self.has_max_vel = [True] * nd
        if nd < 1 or nd > _lib.MAX_DEV:
            raise ValueError(f"{nd} target devices; the C ABI supports 1..{_lib.MAX_DEV}")
        if self.n > _lib.MAX_N or self.k > _lib.MAX_K or self.k < 1:
            raise ValueError(f"n={self.n}, k={self.k} outside the C ABI limits "
                             f"(n<={_lib.MAX_N}, 1<=k<={_lib.MAX_K})")

    @property
    def ndev(self) -> int:
        return len(self.dev_names)

    @property
    def dev_rows(self) -> List[int]:
        return [int(np.sum(m)) for m in self.ctrlr_dof]

    @property
    def k(self) -> int:
        return int(sum(self.dev_rows))

    def flags(self) -> int:
        return ((_lib.USE_G if self.use_g else 0) | (_lib.ADMITTANCE if self.admittance else 0)
                | (_lib.NULLSPACE if self.nullspace else 0))

    def key(self):
        return (self.n, tuple(self.dev_names), tuple(map(tuple, self.ctrlr_dof)),
                tuple(map(tuple, self.joint_ids)), tuple(self.j_idx0), tuple(self.calc_xyz),
                tuple(self.calc_abg), tuple(self.has_max_vel), self.use_g, self.admittance,
                self.nullspace)

    def to_cfg(self, dtype_code: int, max_batch: int, hip_device: int = 0, n_slots: int = 1,
               kernel: int = _lib.KERNEL_AUTO) -> "_lib.Cfg":
        c = _lib.Cfg()
        c.hip_device, c.dtype, c.max_batch, c.n_slots = hip_device, dtype_code, max_batch, n_slots
        c.n, c.ndev, c.flags, c.kernel = self.n, self.ndev, self.flags(), kernel
        for d in range(self.ndev):
            c.dev_rows[d] = self.dev_rows[d]
            for i in range(6):
                c.ctrlr_dof[d][i] = 1 if self.ctrlr_dof[d][i] else 0
            c.calc_xyz[d] = 1 if self.calc_xyz[d] else 0
            c.calc_abg[d] = 1 if self.calc_abg[d] else 0
            mask = 0
            for j in self.joint_ids[d]:
                if not 0 <= int(j) < self.n:
                    raise ValueError(f"device {self.dev_names[d]}: joint id {j} is not a position "
                                     f"in the {self.n}-vector (osc.py:174 would raise IndexError)")
                mask |= 1 << int(j)
            c.joint_mask[d] = mask
            c.j_idx0[d] = int(self.j_idx0[d])
        return c

    def as_oracle_dict(self) -> Dict:
        """The plain-dict form oracle/osc_oracle.generate_batch takes (tests only)."""
        return dict(n=self.n, dev_names=list(self.dev_names), dev_rows=self.dev_rows, ctrlr_dof=self.ctrlr_dof,
                    joint_ids=self.joint_ids, j_idx0=self.j_idx0, has_max_vel=self.has_max_vel,
                    calc_xyz=[bool(x) for x in self.calc_xyz], calc_abg=[bool(x) for x in self.calc_abg],
                    use_g=self.use_g, admittance=self.admittance, nullspace=self.nullspace)

    @classmethod
    def from_devices(cls, devs, robot, use_g: bool = True, admittance: bool = False, nullspace: bool = True,
                     J_idxs=None) -> "OSCLayout":
        """Layout for the target devices `devs` (targets order) of `robot`; J_idxs as returned by
        Robot.get_state(RobotState.J)[1] (robot.py:50-55) or None to derive it from the robot's device order."""
        import numpy as np
        if J_idxs is None:
            J_idxs, row = {}, 0
            for name, dv in robot.sub_devices_dict.items():
                r = int(np.sum(dv.ctrlr_dof))
                J_idxs[name] = np.arange(row, row + r)
                row += r
        return cls(
            n=int(robot.num_joints_total), dev_names=[dv.name for dv in devs],
            ctrlr_dof=[[bool(x) for x in dv.ctrlr_dof] for dv in devs],
            joint_ids=[[int(j) for j in dv.joint_ids_all] for dv in devs],
            j_idx0=[int(J_idxs[dv.name][0]) if len(J_idxs[dv.name]) else 0 for dv in devs],
            calc_xyz=[bool(np.sum(dv.ctrlr_dof_xyz) > 0) for dv in devs],
            calc_abg=[bool(np.sum(dv.ctrlr_dof_abg) > 0) for dv in devs],
            has_max_vel=[dv.max_vel is not None for dv in devs],
            use_g=bool(use_g), admittance=bool(admittance), nullspace=bool(nullspace))

    @classmethod
    def from_dict(cls, d: Dict) -> "OSCLayout":
        nd = len(d["ctrlr_dof"])
        return cls(n=d["n"], dev_names=list(d.get("dev_names", [f"dev{i}" for i in range(nd)])),
                   ctrlr_dof=[list(map(bool, m)) for m in d["ctrlr_dof"]],
                   joint_ids=[list(map(int, j)) for j in d["joint_ids"]], j_idx0=list(d["j_idx0"]),
                   has_max_vel=list(d.get("has_max_vel", [])), use_g=d["use_g"],
                   admittance=d["admittance"], nullspace=d["nullspace"])


def pack_gains(layout: OSCLayout, kp, kv, ko, k, d, max_vel, null_kv=0.0):
    """-> (gains[nb][ndev][12] float64, null_kv[nb] float64, nb).  Inputs broadcast ([ndev],
    [ndev,3], [ndev,2], scalar) or per instance (leading batch axis)."""
    nd = layout.ndev
    kp, kv, ko = (np.asarray(x, dtype=np.float64) for x in (kp, kv, ko))
    k, d, max_vel = (np.asarray(x, dtype=np.float64) for x in (k, d, max_vel))
    null_kv = np.asarray(null_kv, dtype=np.float64)
    lead = [a.shape[0] for a, full in ((kp, 1), (kv, 1), (ko, 1), (k, 2), (d, 2), (max_vel, 2),
                                       (null_kv, 0)) if a.ndim == full + 1]
    nb = max(lead or [1])
    g = np.zeros((nb, nd, GAIN_WORDS))
    g[:, :, 0] = kp
    g[:, :, 1] = kv
    g[:, :, 2] = ko
    g[:, :, 3:6] = k
    g[:, :, 6:9] = d
    g[:, :, 9:11] = max_vel
    g[:, :, 11] = np.asarray(layout.has_max_vel, dtype=np.float64)
    nk = np.broadcast_to(null_kv, (nb,)).astype(np.float64).copy()
    return np.ascontiguousarray(g), nk, nb
