"""3Dconnexion SpaceMouse as an incremental 6-DoF pose (counterpart of /root/reference/irl_control/input_devices/space_mouse.py)."""
import types

import numpy as np


class SpaceMouse:
    """Integrates the device's rate readings into a pose (x, y, z, roll, pitch, yaw): position += increment * reading, yaw and
    pitch count DOWN with their readings, roll up, and every angle is wrapped into (-pi, pi] after its update
    (space_mouse.py:24-34).  `reader` = any callable returning an object with x, y, z, roll, pitch, yaw (a scripted stream, a
    recorded session); None opens the hardware through `pyspacemouse` and raises if there is none (space_mouse.py:7-11)."""

    def __init__(self, origin, increment: float = 0.0015, reader=None):
        if reader is None:
            try:
                import pyspacemouse
            except ImportError as e:
                raise RuntimeError("SpaceMouse: no reader was given and the pyspacemouse package is not installed") from e
            if not pyspacemouse.open():
                raise RuntimeError("Space Mouse not found!")
            reader = pyspacemouse.read
        self._read = reader
        self.increment = increment
        self.state = types.SimpleNamespace(x=origin[0], y=origin[1], z=origin[2], roll=origin[3], pitch=origin[4], yaw=origin[5])

    @staticmethod
    def constrain_angle(angle):
        return np.arctan2(np.sin(angle), np.cos(angle))

    def update_state(self):
        inc, cur, st = self.increment, self._read(), self.state
        st.x += inc * cur.x
        st.y += inc * cur.y
        st.z += inc * cur.z
        st.yaw = self.constrain_angle(st.yaw - inc * cur.yaw)
        st.pitch = self.constrain_angle(st.pitch - inc * cur.pitch)
        st.roll = self.constrain_angle(st.roll + inc * cur.roll)
        return (st.x, st.y, st.z, st.roll, st.pitch, st.yaw)
