"""The tick loops of the reference's two TELEOPERATION callers of the hot path, headless, on injected input streams:

  space_mouse_loop   examples/space_mouse_example.py:106-165 (run_demo)   one 6-DoF pose steers both arms (mirrored grippers
                                                                         around a plate) and the base yaws towards it
  ps_move_loop       examples/ps_move_example.py:85-180 (run)             one controller per arm; the trigger engages an arm
                                                                         (orientation rows controlled, target = controller
                                                                         pose), releasing it holds the arm where it is with
                                                                         the orientation error switched off; circle / triangle
                                                                         close / open the position-controlled gripper

No viewer, no device: `sm` is anything with update_state() (input_devices.SpaceMouse on a scripted reader), `move_states`
the {MoveName: MoveState} records the PS Move collector fills (input_devices.ps_move).  `sim` is whatever was injected; the
tests use FakeSim with ToyDynamics, whose goals follow the mocap bodies the loops move.  Per-tick goldens of both loops are
minted by running the REFERENCE's own loop bodies on the same streams (oracle/make_golden.py: make_teleop_golden).
"""
import numpy as np

from irl_control_amd.input_devices import MoveName
from irl_control_amd.transforms import compose, euler2mat, euler2quat, mat2euler, quat2mat

SPACE_MOUSE_MOCAPS = ("plate", "hand_ur5right", "hand_ur5left")      # scenes/space_mouse_scene.xml:6-14
PS_MOVE_MOCAPS = ("hand_right", "hand_left")                         # what ps_move_example.py:163,167 moves (scenes/world.xml:25-31
                                                                     # has both commented out: the shipped scene lacks them)
GRIP_MIN, GRIP_MAX, GRIP_STEP = 0.0, 0.9, 0.05                       # ps_move_example.py:67-68,77-80


def _pose_targets(tfmat):
    return tfmat[0:3, -1].flatten(), np.array(mat2euler(tfmat[:3, :3]))


def space_mouse_loop(robot, controller, Target, sim, sm, ticks):
    targets = {"ur5right": Target(), "ur5left": Target(), "base": Target()}          # dict order = row / output order
    rec = dict(ctrl=[], pose=[])
    one = [1, 1, 1]
    for _ in range(ticks):
        x, y, z, roll, pitch, yaw = sm.update_state()
        angle = euler2quat(pitch, roll, yaw, axes="rxyz")                            # pitch and roll swapped on purpose (:121)
        plate = compose([x, y, z], quat2mat(angle), one)
        tf_r = plate @ compose([0.05, 0, 0], np.eye(3), one) @ compose([0.15, 0, 0], euler2mat(0, 0, 0), one)
        tf_l = plate @ compose([-0.05, 0, 0], np.eye(3), one) @ compose([-0.15, 0, 0], euler2mat(0, 0, np.pi), one)
        r_xyz, r_ang = _pose_targets(tf_r)
        l_xyz, l_ang = _pose_targets(tf_l)
        sim.data.set_mocap_pos("plate", [x, y, z])
        targets["ur5right"].set_xyz(r_xyz)
        targets["ur5right"].set_abg(r_ang)
        targets["ur5left"].set_xyz(l_xyz)
        targets["ur5left"].set_abg(l_ang)
        targets["base"].set_abg([0, 0, np.arctan2(y, x) - np.pi / 2])
        force_idxs, forces = controller.generate(targets)
        for force_idx, force in zip(force_idxs, forces):
            sim.data.ctrl[force_idx] = force
        sim.data.set_mocap_quat("plate", angle)
        sim.data.set_mocap_pos("hand_ur5right", r_xyz)
        sim.data.set_mocap_quat("hand_ur5right", euler2quat(r_ang[0], r_ang[1], r_ang[2]))
        sim.data.set_mocap_pos("hand_ur5left", l_xyz)
        sim.data.set_mocap_quat("hand_ur5left", euler2quat(l_ang[0], l_ang[1], l_ang[2]))
        sim.step()
        rec["ctrl"].append(np.array(sim.data.ctrl))
        rec["pose"].append((x, y, z, roll, pitch, yaw))
    return dict(ctrl=np.array(rec["ctrl"]), pose=np.array(rec["pose"]))


def update_grip(grip_pos, move_states):
    """One pass of the button poll (ps_move_example.py:72-84): circle closes, triangle opens, clamped to [0, 0.9]."""
    for name in MoveName:
        if move_states[name].get("circle"):
            grip_pos[name] -= GRIP_STEP
        if move_states[name].get("triangle"):
            grip_pos[name] += GRIP_STEP
        grip_pos[name] = min(GRIP_MAX, max(GRIP_MIN, grip_pos[name]))
    return grip_pos


def ps_move_loop(robot, controller, Target, DeviceState, sim, move_states, ticks, advance=None, button_every=4):
    """`advance(tick)` moves the input stream on (called after every rendered tick); the button poll, a 10 Hz thread of its own
    in the reference, runs once every `button_every` ticks here."""
    targets = {"ur5right": Target(), "ur5left": Target(), "base": Target()}
    ur5right, ur5left = robot.get_device("ur5right"), robot.get_device("ur5left")
    grip_pos = {name: 0.0 for name in MoveName}
    one = [1, 1, 1]
    rec = dict(ctrl=[], engaged=[])
    for tick in range(ticks):
        xyz_r, ang_r = move_states[MoveName.RIGHT].get("pos"), move_states[MoveName.RIGHT].get("quat")
        xyz_l, ang_l = move_states[MoveName.LEFT].get("pos"), move_states[MoveName.LEFT].get("quat")
        tf_r = compose(xyz_r, quat2mat(ang_r), one) @ compose([0.0, 0, 0], np.eye(3), one)
        tf_r = tf_r @ compose([0.0, 0, 0], euler2mat(np.pi / 2, 0, np.pi / 2), one)      # gripper facing down the y axis (:110-112)
        tf_l = compose(xyz_l, quat2mat(ang_l), one) @ compose([0.05, 0, 0], np.eye(3), one)
        tf_l = tf_l @ compose([-0.15, 0, 0], euler2mat(0, 0, -np.pi / 2), one)
        r_xyz, r_ang = _pose_targets(tf_r)
        l_xyz, l_ang = _pose_targets(tf_l)
        for dev, name, xyz, ang, tname in ((ur5right, MoveName.RIGHT, r_xyz, r_ang, "ur5right"),
                                           (ur5left, MoveName.LEFT, l_xyz, l_ang, "ur5left")):
            if move_states[name].get("trigger"):
                dev.ctrlr_dof_abg = [True, True, True]
                targets[tname].set_xyz(xyz)
                targets[tname].set_abg(ang)
            else:                                        # released: orientation error off, hold the position it has
                dev.ctrlr_dof_abg = [False, False, False]
                targets[tname].set_xyz(dev.get_state(DeviceState.EE_XYZ))
        force_idxs, forces = controller.generate(targets)
        for force_idx, force in zip(force_idxs, forces):
            sim.data.ctrl[force_idx] = force
            sim.data.ctrl[7] = grip_pos[MoveName.RIGHT]          # gripper position actuators (dual_ur5_grip_pos_ctrl.xml:270,279)
            sim.data.ctrl[14] = grip_pos[MoveName.LEFT]
        sim.data.set_mocap_pos("hand_right", r_xyz)
        sim.data.set_mocap_pos("hand_left", l_xyz)
        move_states[MoveName.RIGHT].set("rumble", sim.data.sensordata[13])
        move_states[MoveName.LEFT].set("rumble", sim.data.sensordata[16])
        sim.step()
        rec["ctrl"].append(np.array(sim.data.ctrl))
        rec["engaged"].append((bool(move_states[MoveName.RIGHT].get("trigger")), bool(move_states[MoveName.LEFT].get("trigger"))))
        if (tick + 1) % button_every == 0:
            update_grip(grip_pos, move_states)
        if advance is not None:
            advance(tick)
    return dict(ctrl=np.array(rec["ctrl"]), engaged=np.array(rec["engaged"]))


# ---- scripted input streams (what the goldens and the tests feed both sides with) ---------------------------------------------
def space_mouse_stream(seed, ticks):
    """Rate readings of a session: slow sinusoids on all six axes (full-scale readings are +-1, the integrator's step 0.0015)."""
    import types
    rng = np.random.default_rng(seed)
    ph, fr = rng.uniform(0, 2 * np.pi, 6), rng.uniform(0.02, 0.09, 6)
    amp = np.array([60.0, 60.0, 40.0, 150.0, 150.0, 250.0])
    state = dict(t=0)

    def read():
        t = state["t"]
        state["t"] += 1
        v = amp * np.sin(fr * t + ph)
        return types.SimpleNamespace(x=v[0], y=v[1], z=v[2], roll=v[3], pitch=v[4], yaw=v[5])
    return read


def ps_move_script(seed, ticks):
    """Per tick and controller: pos[3], quat[4], trigger, circle, triangle -- smooth paths inside the scene ranges of the two
    trackers, the triggers pressed and released a few times, bursts of circle / triangle."""
    from irl_control_amd.transforms import normalized_vector
    rng = np.random.default_rng(seed)
    out = []
    centre = {MoveName.RIGHT: np.array([0.3, 0.5, 0.3]), MoveName.LEFT: np.array([-0.3, 0.5, 0.3])}
    ph = {n: rng.uniform(0, 2 * np.pi, 7) for n in MoveName}
    for t in range(ticks):
        row = {}
        for i, n in enumerate(MoveName):
            pos = centre[n] + np.array([0.25, 0.25, 0.15]) * np.sin(0.03 * t * np.array([1.0, 0.7, 1.3]) + ph[n][:3])
            quat = normalized_vector(np.array([1.0, 0.0, 0.0, 0.0]) + 0.35 * np.sin(0.02 * t * np.array([1.0, 1.4, 0.6, 0.9]) + ph[n][3:]))
            trig = ((t + 37 * i) // 45) % 3 != 1                     # engaged, released, engaged, ...
            row[n] = dict(pos=pos, quat=quat, trigger=bool(trig), circle=bool(60 + 20 * i <= t < 90 + 20 * i),
                          triangle=bool(20 * i <= t < 40 + 20 * i))
        out.append(row)
    return out


def apply_script_row(move_states, row):
    for n, r in row.items():
        for k, v in r.items():
            move_states[n].set(k, v)
