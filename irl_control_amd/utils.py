"""Import-path parity with the reference: ``from irl_control.utils import Target, ControllerConfig``."""
from .targets import ControllerConfig, Target  # noqa: F401
