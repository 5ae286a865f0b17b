"""Quaternion / Euler helpers used on the OSC path (w,x,y,z quaternions, static 'sxyz' Euler).

The reference takes these from the third-party ``transforms3d`` package
(call sites: /root/reference/irl_control/osc.py:4-7,115-117 and
/root/reference/irl_control/utils.py:3,14-15,30,33,53,57,67).  That package is neither vendored by
the reference nor installed in this image, so these are restatements of its published formulas for
the conventions the reference uses (axes='sxyz' on the OSC path; 'rxyz' in the space-mouse caller).  They are cross-checked against
``scipy.spatial.transform.Rotation`` in tests/test_transforms.py.  PARITY UNPINNED with respect to
transforms3d itself (SURVEY.md §8 a6').
"""
import math

import numpy as np

_EPS4 = float(np.finfo(np.float64).eps) * 4.0


def qmult(q1, q2):
    """Hamilton product, no normalisation (transforms3d.derivations.quaternions.qmult)."""
    w1, x1, y1, z1 = q1
    w2, x2, y2, z2 = q2
    w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    y = w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2
    z = w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2
    return w, x, y, z


def qconjugate(q):
    return np.array(q, dtype=np.float64) * np.array([1.0, -1.0, -1.0, -1.0])


def normalized_vector(v):
    v = np.asarray(v, dtype=np.float64)
    return v / math.sqrt(float((v ** 2).sum()))


def quat2mat(q):
    w, x, y, z = q
    Nq = w * w + x * x + y * y + z * z
    if Nq < np.finfo(np.float64).eps:
        return np.eye(3)
    s = 2.0 / Nq
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY],
                     [xY + wZ, 1.0 - (xX + zZ), yZ - wX],
                     [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])


def mat2euler(M):
    """Static-frame x-y-z ('sxyz') Euler angles of a rotation matrix."""
    cy = math.sqrt(M[0, 0] * M[0, 0] + M[1, 0] * M[1, 0])
    if cy > _EPS4:
        ax = math.atan2(M[2, 1], M[2, 2])
        ay = math.atan2(-M[2, 0], cy)
        az = math.atan2(M[1, 0], M[0, 0])
    else:
        ax = math.atan2(-M[1, 2], M[1, 1])
        ay = math.atan2(-M[2, 0], cy)
        az = 0.0
    return ax, ay, az


def quat2euler(q):
    return mat2euler(quat2mat(q))


_AXIS = {"x": 0, "y": 1, "z": 2}


def _axes_sequence(axes):
    """'sxyz' / 'rzyx' / ... -> the three (axis index, angle slot) rotations in the order they are applied about FIXED axes.
    transforms3d's naming (call sites: examples/space_mouse_example.py:59,121): first letter s = static frame (rotate about
    fixed x, then fixed y, then fixed z for 'sxyz'), r = rotating frame (about x, then the NEW y, then the NEW z for
    'rxyz'), which is the same rotation as the static sequence taken in the opposite order."""
    if len(axes) != 4 or axes[0] not in "sr" or any(c not in _AXIS for c in axes[1:]) or axes[1] == axes[2] or axes[2] == axes[3]:
        raise ValueError(f"axes={axes!r}")
    seq = [(_AXIS[c], slot) for slot, c in enumerate(axes[1:])]
    return seq if axes[0] == "s" else seq[::-1]


def _axis_quat(axis, angle):
    q = np.zeros(4)
    q[0] = math.cos(angle / 2.0)
    q[1 + axis] = math.sin(angle / 2.0)
    return q


def euler2quat(ai, aj, ak, axes="sxyz"):
    """Euler angles -> (w,x,y,z); 'sxyz' (every call on the OSC path) by the closed form, the other conventions of
    transforms3d.euler by composing the three elementary rotations."""
    if axes != "sxyz":
        q = np.array([1.0, 0.0, 0.0, 0.0])
        for axis, slot in _axes_sequence(axes):
            q = np.array(qmult(_axis_quat(axis, (ai, aj, ak)[slot]), q))      # later rotations about fixed axes multiply from the left
        return q
    ai, aj, ak = ai / 2.0, aj / 2.0, ak / 2.0
    ci, si = math.cos(ai), math.sin(ai)
    cj, sj = math.cos(aj), math.sin(aj)
    ck, sk = math.cos(ak), math.sin(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return np.array([cj * cc + sj * ss, cj * sc - sj * cs, cj * ss + sj * cc, cj * cs - sj * sc])


def euler2mat(ai, aj, ak, axes="sxyz"):
    """Euler angles -> rotation matrix (transforms3d.euler.euler2mat); 'sxyz': Rz(ak) Ry(aj) Rx(ai)."""
    if axes != "sxyz":
        return quat2mat(euler2quat(ai, aj, ak, axes))
    si, sj, sk = math.sin(ai), math.sin(aj), math.sin(ak)
    ci, cj, ck = math.cos(ai), math.cos(aj), math.cos(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return np.array([[cj * ck, sj * sc - cs, sj * cc + ss],
                     [cj * sk, sj * ss + cc, sj * cs - sc],
                     [-sj, cj * si, cj * ci]])


def compose(T, R, Z):
    """4 x 4 affine from translation, rotation matrix and zooms (transforms3d.affines.compose, no shear)."""
    A = np.eye(4)
    A[:3, :3] = np.asarray(R, dtype=np.float64) * np.asarray(Z, dtype=np.float64)[None, :]
    A[:3, 3] = T
    return A
