"""Host-side state keeping of the two teleoperation input devices (irl_control_amd/input_devices/, counterpart of
/root/reference/irl_control/input_devices/): no hardware, no GPU."""
import types

import numpy as np
import pytest

from irl_control_amd import fakesim
from irl_control_amd.input_devices import MoveName, MoveState, PSMoveInterface, SpaceMouse, move_orientation, rumble_level, tracker_to_sim
from irl_control_amd.transforms import euler2quat, quat2euler


def test_space_mouse_integrates_rates_and_wraps_angles():
    """space_mouse.py:24-34: position += inc * reading; yaw and pitch count DOWN, roll up; every angle wrapped into (-pi, pi]."""
    reads = iter([types.SimpleNamespace(x=1.0, y=-2.0, z=0.5, roll=100.0, pitch=200.0, yaw=-300.0)] * 3)
    sm = SpaceMouse([0.0, 0.5, 0.5, 3.0, -3.0, 3.1], increment=0.01, reader=lambda: next(reads))
    x, y, z, roll, pitch, yaw = sm.update_state()
    assert np.allclose([x, y, z], [0.01, 0.48, 0.505])
    assert np.isclose(roll, 3.0 + 1.0 - 2 * np.pi) and np.isclose(pitch, -3.0 - 2.0 + 2 * np.pi) and np.isclose(yaw, 3.1 + 3.0 - 2 * np.pi)
    sm.update_state()
    assert np.isclose(sm.state.x, 0.02) and -np.pi < sm.state.roll <= np.pi


def test_space_mouse_without_reader_or_driver_fails_loudly(monkeypatch):
    import sys
    monkeypatch.setitem(sys.modules, "pyspacemouse", None)
    with pytest.raises(RuntimeError, match="pyspacemouse"):
        SpaceMouse([0, 0, 0, 0, 0, 0])


def test_ps_move_tracker_ranges_orientation_and_rumble():
    """ps_move.py:96-114,148-153,170-192: tracker x -> scene x, sphere radius -> scene y, minus tracker y -> scene z, clamped;
    the orientation keeps the first Euler angle and feeds the second into the third slot; rumble ramps 0..130."""
    assert np.allclose(tracker_to_sim(MoveName.RIGHT, 150, 400, 12), [0.7, 0.9, 0.01])
    assert np.allclose(tracker_to_sim(MoveName.RIGHT, 375, 20, 70), [-0.2, 0.0, 0.5])
    assert np.allclose(tracker_to_sim(MoveName.LEFT, 1000, -1000, 0), [-0.7, 0.9, 0.5])          # clamped on all three axes
    assert np.allclose(tracker_to_sim(MoveName.LEFT, 487.5, 210, 41), [-0.25, 0.45, 0.255])
    q = euler2quat(0.4, -0.3, 0.9)
    e = quat2euler(move_orientation(q))
    assert np.allclose(e, [0.4, 0.0, -0.3])
    assert rumble_level(0.0) == 0 and rumble_level(-0.1) == 0 and rumble_level(-0.4) == 65 and rumble_level(-5.0) == 130


def test_ps_move_interface_polls_an_injected_source_and_refuses_without_one():
    src = iter([{MoveName.RIGHT: dict(trigger_value=200, x=262.5, y=210, radius=41, quat=[1, 0, 0, 0], circle=True),
                 MoveName.LEFT: dict(trigger_value=3, tracking=False, quat=euler2quat(0.2, 0.1, 0.0))}])
    itf = PSMoveInterface(source=src)
    st = itf.poll()
    assert st[MoveName.RIGHT].get("trigger") and st[MoveName.RIGHT].get("circle") and not st[MoveName.RIGHT].get("triangle")
    assert np.allclose(st[MoveName.RIGHT].get("pos"), [0.25, 0.45, 0.255])
    assert not st[MoveName.LEFT].get("trigger") and np.array_equal(st[MoveName.LEFT].get("pos"), np.zeros(3))   # not tracking: pos kept
    assert isinstance(st[MoveName.LEFT], MoveState)
    with pytest.raises((RuntimeError, NotImplementedError)):
        PSMoveInterface()


def test_fakesim_scene_mocap_bodies():
    """Extra mocap bodies of a scene (space_mouse_scene.xml:6-14) sit behind the robot's bodies: the Device index maps do not move."""
    plain, scene = fakesim.FakeSim(), fakesim.FakeSim(mocap_names=("plate", "hand_ur5right"))
    assert scene.model.nbody == plain.model.nbody + 2 and scene.model.nv == plain.model.nv
    assert scene.model.body_name2id("ur_EE_ur5left") == plain.model.body_name2id("ur_EE_ur5left")
    scene.data.set_mocap_pos("plate", [1, 2, 3])
    scene.data.set_mocap_quat("hand_ur5right", [0, 1, 0, 0])
    assert np.array_equal(scene.data.get_body_xpos("plate"), [1, 2, 3]) and np.array_equal(scene.data.get_body_xquat("hand_ur5right"), [0, 1, 0, 0])
    with pytest.raises(ValueError):
        plain.data.set_mocap_pos("plate", [0, 0, 0])
