from irl_control_amd import transforms as _t
from irl_control_amd.transforms import quat2mat, mat2euler  # noqa: F401


def _only_sxyz(axes):
    if axes not in ("sxyz",):
        raise NotImplementedError(f"transforms3d shim: axes={axes!r}")


def euler2quat(ai, aj, ak, axes="sxyz"):
    return _t.euler2quat(ai, aj, ak, axes)


def euler2mat(ai, aj, ak, axes="sxyz"):
    return _t.euler2mat(ai, aj, ak, axes)


def quat2euler(q, axes="sxyz"):
    if axes == "rxyz":          # only examples/insertion_task.py:19 (an unused module constant): intrinsic x-y-z via scipy
        from scipy.spatial.transform import Rotation
        w, x, y, z = q
        return tuple(Rotation.from_quat([x, y, z, w]).as_euler("XYZ"))
    _only_sxyz(axes)
    return _t.quat2euler(q)
