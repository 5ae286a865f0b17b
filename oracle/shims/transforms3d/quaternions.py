import numpy as _np

from irl_control_amd.transforms import qconjugate  # noqa: F401
from irl_control_amd import transforms as _t


def qmult(q1, q2):
    """transforms3d.quaternions.qmult returns an array (the derivations variant returns a tuple)."""
    return _np.array(_t.qmult(q1, q2))
