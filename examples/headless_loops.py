"""The tick loops of the reference's two hot-path callers, headless, written against the call surface the reference
and this package share, so that ONE piece of code drives both: the reference itself when the per-tick goldens are
minted (oracle/make_golden.py, build container only) and the HIP path in the examples and the GPU tests.

  gain_test_loop   examples/gain_test.py:98-175   targets right / left / base, waypoint switching at 0.1 m
  admit_test_loop  examples/admit_test.py:43-80   two arms, admittance on, left arm abg = [0, -pi/2, 0], a push on
                                                  the left gripper during a window of ticks

No viewer, no timers, no MuJoCo: `sim` is whatever was injected (an MjSim works as it is; the tests use FakeSim with
fakesim.ToyDynamics, which both loops feed with the current goals).  Each loop returns the per-tick record
{"forces": [ticks, n_act], "idxs": ..., ...}.
"""
import numpy as np

THRESHOLD_EE = 0.1                                         # examples/gain_test.py:118
EE_BODY = {"ur5right": "ur_EE_ur5right", "ur5left": "ur_EE_ur5left"}


def gain_test_waypoints():
    """The back-and-forth path of the reference demo (examples/gain_test.py:80-96)."""
    return np.array([[0.8, 0.6, 0.7], [0.8, -0.6, 0.7]]), np.array([[-0.5, -0.5, 0.5]])


def figure_eight_waypoints():
    """Closed polyline through six corner points per arm, four interpolated points per edge, mirrored for the right arm."""
    corners = np.array([[-0.8, -0.4, 0.5], [-0.9, -0.35, 0.7], [-0.9, -0.2, 0.5],
                        [-0.9, -0.6, 0.2], [-0.7, -0.7, 0.2], [-0.4, -0.8, 0.3]])
    closed = np.vstack([corners, corners[:1]])
    left = np.vstack([np.linspace(closed[i], closed[i + 1], 5) for i in range(len(corners))])
    right = left.copy()
    right[:, :2] *= -1
    return right, left


def gain_test_loop(robot, controller, Target, DeviceState, sim, ticks, waypoints=None, dyn=None):
    right_wps, left_wps = waypoints if waypoints is not None else gain_test_waypoints()
    targets = {"ur5right": Target(), "ur5left": Target(), "base": Target()}      # dict order = row / output order
    ur5right, ur5left = robot.get_device("ur5right"), robot.get_device("ur5left")
    ri = li = 0
    rec = dict(forces=[], wp=[], err=[], idxs=None)
    for _ in range(ticks):
        targets["ur5right"].set_xyz(right_wps[ri])
        targets["ur5left"].set_xyz(left_wps[li])
        force_idxs, forces = controller.generate(targets)
        for force_idx, force in zip(force_idxs, forces):
            sim.data.ctrl[force_idx] = force
        err_r = np.linalg.norm(ur5right.get_state(DeviceState.EE_XYZ) - targets["ur5right"].get_xyz())
        err_l = np.linalg.norm(ur5left.get_state(DeviceState.EE_XYZ) - targets["ur5left"].get_xyz())
        if err_r < THRESHOLD_EE:
            ri = ri + 1 if ri < right_wps.shape[0] - 1 else 0
        if err_l < THRESHOLD_EE:
            li = li + 1 if li < left_wps.shape[0] - 1 else 0
        rec["forces"].append(np.concatenate([np.asarray(f, dtype=np.float64) for f in forces]))
        rec["wp"].append((ri, li))
        rec["err"].append((err_r, err_l))
        if rec["idxs"] is None:
            rec["idxs"] = [np.asarray(x).tolist() for x in force_idxs]
        sim.data.set_mocap_pos("target_red", right_wps[ri])
        sim.data.set_mocap_pos("target_blue", left_wps[li])
        if dyn is not None:
            dyn.goal_xyz = {EE_BODY["ur5right"]: right_wps[ri], EE_BODY["ur5left"]: left_wps[li]}
        sim.step()
    rec["forces"] = np.array(rec["forces"]); rec["wp"] = np.array(rec["wp"]); rec["err"] = np.array(rec["err"])
    return rec


def admit_test_loop(robot, controller, Target, DeviceState, sim, ticks, push_window=(3000, 5000), dyn=None,
                    push=(20.0, 0.0, 0.0, 0.0, 0.0, 0.0)):
    body_id = sim.model.body_name2id("left_outer_knuckle_ur5left")
    targets = {"ur5right": Target(), "ur5left": Target()}
    right_wp, left_wp = np.array([0.3, 0.46432, 0.5]), np.array([-0.3, 0.46432, 0.5])       # admit_test.py:35-41
    rec = dict(forces=[], idxs=None)
    count = 0
    for _ in range(ticks):
        count += 1
        targets["ur5right"].set_xyz(right_wp)
        targets["ur5left"].set_xyz(left_wp)
        targets["ur5left"].set_abg(np.array([0, -1 * np.pi / 2, 0]))
        sim.data.set_mocap_pos("target_red", right_wp)
        sim.data.set_mocap_pos("target_blue", left_wp)
        force_idxs, forces = controller.generate(targets)
        for force_idx, force in zip(force_idxs, forces):
            sim.data.ctrl[force_idx] = force
        sim.data.xfrc_applied[body_id] = [0, 0, 0, 0, 0, 0]
        if push_window[0] < count < push_window[1]:
            sim.data.xfrc_applied[body_id] = list(push)
        rec["forces"].append(np.concatenate([np.asarray(f, dtype=np.float64) for f in forces]))
        if rec["idxs"] is None:
            rec["idxs"] = [np.asarray(x).tolist() for x in force_idxs]
        if dyn is not None:
            dyn.goal_xyz = {EE_BODY["ur5right"]: right_wp, EE_BODY["ur5left"]: left_wp}
            dyn.goal_quat = {EE_BODY["ur5left"]: targets["ur5left"].get_quat()}
        sim.step()
    rec["forces"] = np.array(rec["forces"])
    return rec


def force_test_loop(robot, controller, Target, DeviceState, sim, ticks, dyn=None, threshold_ee=0.01):
    """examples/force_test.py:57-127: admittance controller, the left arm walks a line of waypoints (threshold 1 cm)
    with orientation target abg = [0, 0, -pi/2], and the left F/T force is read back after every step (the reference
    logs it to data.csv; `mj_inverse` refreshes the sensors there, here `sim.inverse()` is called when the backend has it)."""
    right_wps = np.array([[0.3, 0.46432, 0.36243]])
    left_wps = np.array([[-0.3, y, 0.5] for y in (0.46432, 0.5, 0.55, 0.575, 0.6, 0.625, 0.65, 0.675, 0.7, 0.75)])
    targets = {"ur5right": Target(), "ur5left": Target()}
    ur5left = robot.sub_devices_dict["ur5left"]
    ri = li = 0
    rec = dict(forces=[], wp=[], ft=[], idxs=None)
    for _ in range(ticks):
        targets["ur5right"].set_xyz(right_wps[ri])
        targets["ur5left"].set_xyz(left_wps[li])
        targets["ur5left"].set_abg(np.array([0, 0, -1 * np.pi / 2]))
        sim.data.set_mocap_pos("target_red", right_wps[ri])
        sim.data.set_mocap_pos("target_blue", left_wps[li])
        force_idxs, forces = controller.generate(targets)
        for force_idx, force in zip(force_idxs, forces):
            sim.data.ctrl[force_idx] = force
        err_l = np.linalg.norm(ur5left.get_state(DeviceState.EE_XYZ) - targets["ur5left"].get_xyz())
        if err_l < threshold_ee:
            li = li + 1 if li < left_wps.shape[0] - 1 else 0
        if dyn is not None:
            dyn.goal_xyz = {EE_BODY["ur5right"]: right_wps[ri], EE_BODY["ur5left"]: left_wps[li]}
            dyn.goal_quat = {EE_BODY["ur5left"]: targets["ur5left"].get_quat()}
        sim.step()
        if hasattr(sim, "inverse"):
            sim.inverse()
        rec["forces"].append(np.concatenate([np.asarray(f, dtype=np.float64) for f in forces]))
        rec["wp"].append((ri, li))
        rec["ft"].append(np.array(ur5left.get_state(DeviceState.FORCE), dtype=np.float64))
        if rec["idxs"] is None:
            rec["idxs"] = [np.asarray(x).tolist() for x in force_idxs]
    rec["forces"] = np.array(rec["forces"]); rec["wp"] = np.array(rec["wp"]); rec["ft"] = np.array(rec["ft"])
    return rec
