"""MI355X-native batched operational-space controller with the irl_control API surface.

``import irl_control_amd as irl_control`` gives the names the reference package exports
(/root/reference/irl_control/__init__.py:1-5): Device, Robot, OSC, MujocoApp — plus the batched
entry point ``BatchedOSC`` and the headless ``FakeSim`` backend.
"""
from .version import version as __version__  # noqa: F401
from .device import Device, DeviceState  # noqa: F401
from .robot import Robot, RobotState  # noqa: F401
from .targets import ControllerConfig, Target  # noqa: F401
from .layout import OSCLayout  # noqa: F401
from .batched import BatchedOSC  # noqa: F401
from .osc import OSC  # noqa: F401
from .mujoco_app import MujocoApp  # noqa: F401
from .fakesim import FakeSim  # noqa: F401
from .action_sequence import ActionSequenceRunner  # noqa: F401
