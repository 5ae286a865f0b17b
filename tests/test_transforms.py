"""The restated transforms3d 'sxyz' formulas vs scipy (independent implementation)."""
import numpy as np
from scipy.spatial.transform import Rotation

from irl_control_amd import transforms as tf


def test_quat2euler_matches_scipy():
    rng = np.random.default_rng(0)
    q = rng.normal(size=(5000, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    ours = np.array([tf.quat2euler(x) for x in q])
    ref = Rotation.from_quat(q, scalar_first=True).as_euler("xyz")
    d = np.abs(ours - ref)
    d = np.minimum(d, 2 * np.pi - d)
    assert d.max() < 1e-12


def test_euler2quat_matches_scipy_up_to_sign():
    rng = np.random.default_rng(1)
    e = rng.uniform(-np.pi, np.pi, size=(5000, 3))
    ours = np.array([tf.euler2quat(*x) for x in e])
    ref = Rotation.from_euler("xyz", e).as_quat(scalar_first=True)
    d = np.minimum(np.abs(ours - ref).max(axis=1), np.abs(ours + ref).max(axis=1))
    assert d.max() < 1e-14


def test_roundtrip_and_unnormalised_quat():
    rng = np.random.default_rng(2)
    for _ in range(200):
        e = rng.uniform(-1.5, 1.5, size=3)
        q = tf.euler2quat(*e)
        assert np.allclose(tf.quat2euler(q), e, atol=1e-12)
        assert np.allclose(tf.quat2euler(q * 3.7), e, atol=1e-12)   # quat2mat divides by |q|^2


def test_gimbal_branch():
    # ay = +pi/2 exactly: cy == 0 -> az forced to 0 (transforms3d mat2euler degenerate branch)
    q = tf.euler2quat(0.3, np.pi / 2, 0.0)
    ax, ay, az = tf.quat2euler(q)
    assert az == 0.0 or abs(az) < 1e-6
    assert abs(ay - np.pi / 2) < 1e-7


def test_qmult_is_hamilton_product():
    rng = np.random.default_rng(3)
    a, b = rng.normal(size=4), rng.normal(size=4)
    r = Rotation.from_quat(a / np.linalg.norm(a), scalar_first=True) * \
        Rotation.from_quat(b / np.linalg.norm(b), scalar_first=True)
    ours = np.array(tf.qmult(a, b)) / (np.linalg.norm(a) * np.linalg.norm(b))
    ref = r.as_quat(scalar_first=True)
    assert min(np.abs(ours - ref).max(), np.abs(ours + ref).max()) < 1e-14


def test_euler_axes_conventions_vs_scipy():
    """euler2quat / euler2mat for the other conventions of transforms3d.euler ('rxyz' is what examples/space_mouse_example.py:59,121
    asks for): static = extrinsic (SciPy lower case), rotating = intrinsic (SciPy upper case)."""
    from scipy.spatial.transform import Rotation
    from irl_control_amd import transforms as t
    rng = np.random.default_rng(11)
    for axes in ("sxyz", "rxyz", "szyx", "rzyx", "szxz", "rxzx", "syxz", "ryzx"):
        for _ in range(50):
            a = rng.uniform(-3, 3, 3)
            R = Rotation.from_euler(axes[1:] if axes[0] == "s" else axes[1:].upper(), a).as_matrix()
            assert np.abs(t.quat2mat(t.euler2quat(*a, axes=axes)) - R).max() < 1e-14
            assert np.abs(t.euler2mat(*a, axes=axes) - R).max() < 1e-14
    assert np.array_equal(t.euler2quat(0.3, -0.2, 0.1), t.euler2quat(0.3, -0.2, 0.1, axes="sxyz"))
    import pytest
    with pytest.raises(ValueError):
        t.euler2quat(0, 0, 0, axes="sxxz")
