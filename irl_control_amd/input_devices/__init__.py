"""Host-side state keeping of the two teleoperation input devices of the reference (irl_control/input_devices/): what turns
raw device readings into the poses the caller loops hand to ``OSC.generate``.  The hardware libraries (``pyspacemouse``,
``psmove``) are only imported when no reader is injected; neither exists in the build image or on the GPU box."""
from .ps_move import MoveName, MoveState, PSMoveInterface, move_orientation, rumble_level, tracker_to_sim  # noqa: F401
from .space_mouse import SpaceMouse  # noqa: F401
