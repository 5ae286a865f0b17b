"""Stand-in for the `transforms3d` package (absent from the image and from the reference tree, requirements.in:3, unpinned) for
oracle/make_golden.py ONLY.

Independent of the product on purpose: nothing here imports irl_control_amd.  The rotation conversions go
through SciPy's `Rotation` (another code base, another algorithm: it orthonormalises matrices and composes elementary rotations),
with transforms3d's published conventions stated explicitly where SciPy has none of its own:

  * quaternions are (w, x, y, z); quat2mat of a quaternion with |q|^2 < eps is the identity (transforms3d.quaternions.quat2mat);
  * mat2euler('sxyz') at gimbal lock -- cy = sqrt(M00^2 + M10^2) <= 4 eps -- returns (atan2(-M12, M11), atan2(-M20, cy), 0)
    (transforms3d.euler.mat2euler, `_EPS4` branch).  SciPy declares gimbal lock three hundred million times earlier (|cos| < 1e-7)
    and then splits the two coupled angles its own way; and this close to the lock two of the three angles are quotients of matrix
    entries of size cy, so WHICH rounding the matrix carries is amplified by 1 / cy (1e9 in the k13_gimbal fixture, which exists to pin
    the reference's branch there).  For cy < 1e-6 quat2euler therefore is transforms3d's published arithmetic end to end -- quat2mat's
    formula, then mat2euler's two branches -- restated here (standalone: a third statement next to the product's and the oracle's);
    everywhere else SciPy does the work.

So the only thing the goldens and the product share is the reference itself; the product's own restatement (irl_control_amd/transforms.py,
the kernels) is compared with these numbers by the golden tests, and with SciPy directly by tests/test_transforms.py.
"""
import math
import warnings

import numpy as np
from scipy.spatial.transform import Rotation

_EPS = float(np.finfo(np.float64).eps)


def _rot(q):
    w, x, y, z = (float(v) for v in q)
    return Rotation.from_quat([x, y, z, w])


def _scipy_axes(axes):
    """transforms3d 'sxyz' (static frame, first letter s) = SciPy extrinsic 'xyz' (lower case); 'rxyz' = intrinsic 'XYZ'."""
    if len(axes) != 4 or axes[0] not in "sr" or any(c not in "xyz" for c in axes[1:]):
        raise ValueError(f"axes={axes!r}")
    return axes[1:] if axes[0] == "s" else axes[1:].upper()


def quat2mat(q):
    w, x, y, z = (float(v) for v in q)
    if w * w + x * x + y * y + z * z < _EPS:
        return np.eye(3)
    return _rot(q).as_matrix()


def mat2euler(M, axes="sxyz"):
    M = np.asarray(M, dtype=np.float64)
    if axes == "sxyz":
        cy = math.sqrt(M[0, 0] * M[0, 0] + M[1, 0] * M[1, 0])
        if not cy > 4.0 * _EPS:                      # gimbal lock: transforms3d's convention, stated explicitly
            return math.atan2(-M[1, 2], M[1, 1]), math.atan2(-M[2, 0], cy), 0.0
        if cy < 1e-6:                                # not yet gimbal lock for transforms3d, already for SciPy: transforms3d's regular branch
            return math.atan2(M[2, 1], M[2, 2]), math.atan2(-M[2, 0], cy), math.atan2(M[1, 0], M[0, 0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = Rotation.from_matrix(M).as_euler(_scipy_axes(axes))
    return float(a[0]), float(a[1]), float(a[2])


def _quat2mat_published(q):
    """transforms3d.quaternions.quat2mat, the published formula (used near gimbal lock only, see the header)."""
    w, x, y, z = (float(v) for v in q)
    Nq = w * w + x * x + y * y + z * z
    if Nq < _EPS:
        return np.eye(3)
    s = 2.0 / Nq
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY], [xY + wZ, 1.0 - (xX + zZ), yZ - wX], [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])


def quat2euler(q, axes="sxyz"):
    M = quat2mat(q)
    if axes == "sxyz" and math.sqrt(M[0, 0] * M[0, 0] + M[1, 0] * M[1, 0]) < 1e-6:
        M = _quat2mat_published(q)
    return mat2euler(M, axes)


def euler2quat(ai, aj, ak, axes="sxyz"):
    x, y, z, w = Rotation.from_euler(_scipy_axes(axes), [ai, aj, ak]).as_quat()
    return np.array([w, x, y, z])


def euler2mat(ai, aj, ak, axes="sxyz"):
    return Rotation.from_euler(_scipy_axes(axes), [ai, aj, ak]).as_matrix()


def qmult(q1, q2):
    """Hamilton product (the definition; transforms3d.quaternions.qmult returns an array)."""
    w1, v1 = float(q1[0]), np.asarray(q1[1:], dtype=np.float64)
    w2, v2 = float(q2[0]), np.asarray(q2[1:], dtype=np.float64)
    return np.concatenate([[w1 * w2 - float(v1 @ v2)], w1 * v2 + w2 * v1 + np.cross(v1, v2)])


def qconjugate(q):
    q = np.array(q, dtype=np.float64)
    q[1:] = -q[1:]
    return q


def normalized_vector(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def compose(T, R, Z):
    """4 x 4 affine from translation, rotation matrix and zooms (transforms3d.affines.compose without shear)."""
    A = np.eye(4)
    A[:3, :3] = np.asarray(R, dtype=np.float64) @ np.diag(np.asarray(Z, dtype=np.float64))
    A[:3, 3] = np.asarray(T, dtype=np.float64)
    return A
