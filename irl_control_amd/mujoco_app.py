"""``MujocoApp``: loads the device / robot / gain tables and builds Device and Robot objects
(API of /root/reference/irl_control/mujoco_app.py:10-62).

The simulator is a backend: pass ``sim=`` (anything exposing the member set listed in
irl_control_amd/fakesim.py, e.g. ``FakeSim()``, a mujoco_py ``MjSim`` or ``mujoco_backend.MujocoSim``).
Without ``sim`` the scene file is loaded with the official ``mujoco`` bindings (through the MujocoSim
adapter) when they are installed, else with mujoco_py exactly like the reference.
"""
import os
import time
from typing import Dict

import numpy as np
import yaml

from .device import Device
from .robot import Robot

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))


class MujocoApp():
    def __init__(self, robot_config_file: str = None, scene_file: str = None, use_sim: bool = True,
                 sim=None):
        cfg_path = robot_config_file if os.path.isabs(robot_config_file or "") else \
            os.path.join(_PKG_DIR, "robot_configs", robot_config_file)
        with open(cfg_path, 'r') as f:
            self.config = yaml.safe_load(f)
        if sim is None:
            if scene_file is None:
                raise ValueError("MujocoApp needs either sim= (an MjSim-like object) or scene_file=")
            scene = scene_file if os.path.isabs(scene_file) else os.path.join(_PKG_DIR, "scenes", scene_file)
            if not os.path.exists(scene):
                # the reference resolves a relative scene against its own scenes/ directory (mujoco_app.py:14-16); this package
                # ships no MJCF scenes or meshes (out of scope), so a relative name cannot be found here
                raise FileNotFoundError(
                    f"scene file {scene!r} not found: irl_control_amd ships no scenes/ directory -- pass an absolute path to an MJCF "
                    f"file (e.g. <irl_control checkout>/irl_control/scenes/{os.path.basename(scene_file)}) or inject sim=")
            try:
                import mujoco  # noqa: F401
                from .mujoco_backend import MujocoSim
                self.sim = MujocoSim.from_xml_path(scene)
                self.model = self.sim.model
            except ImportError:
                try:
                    from mujoco_py import MjSim, load_model_from_path
                except ImportError as e:
                    raise ImportError("no simulator backend: pass sim=FakeSim() (or any MjSim-like object); neither "
                                      "mujoco nor mujoco_py is installed") from e
                self.model = load_model_from_path(scene)
                self.sim = MjSim(self.model)
        else:
            self.sim = sim
            self.model = sim.model
        self.devices = np.array([Device(dev, self.model, self.sim, use_sim)
                                 for dev in self.config['devices']])
        self.create_robot_devices(self.config['robots'], use_sim)
        self.controller_configs = self.config['controller_configs']
        self.timer_running = False

    def create_robot_devices(self, robot_yml: Dict, use_sim: bool):
        """Replace the devices claimed by a robot entry with one Robot object each."""
        robots, claimed = [], []
        for rbt in robot_yml:
            ids = list(rbt['device_ids'])
            claimed += ids
            robots.append(Robot(list(self.devices[ids]), rbt['name'], self.sim, use_sim))
        free = [d for i, d in enumerate(self.devices) if i not in set(claimed)]
        self.devices = np.array(free + robots, dtype=object)

    def sleep_for(self, sleep_time: float):
        assert self.timer_running == False  # noqa: E712
        self.timer_running = True
        time.sleep(sleep_time)
        self.timer_running = False

    def get_robot(self, robot_name: str) -> Robot:
        for device in self.devices:
            if type(device) == Robot and device.name == robot_name:
                return device

    def get_controller_config(self, name: str) -> Dict:
        for entry in self.config['controller_configs']:
            if entry['name'] == name:
                return entry

    def set_free_joint_qpos(self, free_joint_name, quat=None, pos=None):
        jnt_id = self.sim.model.joint_name2id(free_joint_name)
        offset = self.sim.model.jnt_qposadr[jnt_id]
        if quat is not None:
            self.sim.data.qpos[offset + 3: offset + 7] = quat
        if pos is not None:
            self.sim.data.qpos[offset: offset + 3] = pos
