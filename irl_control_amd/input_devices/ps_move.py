"""PS Move controllers as shared state records (counterpart of /root/reference/irl_control/input_devices/ps_move.py).

The caller loop (examples/ps_move_example.py:96-180) only ever reads `move_states[MoveName.X].get(key)` for
pos / quat / trigger / circle / triangle and writes `set('rumble', v)`; what fills those records from the hardware is the
collector below.  Its arithmetic -- tracker pixel ranges to scene coordinates, the orientation with the pitch taken out,
the rumble ramp -- is in plain functions so that it can be driven (and tested) without a controller."""
from enum import Enum
from typing import Any, Dict

import numpy as np

from ..transforms import euler2quat, quat2euler


class MoveName(Enum):
    RIGHT = 0
    LEFT = 1


class MoveState:
    """pos [3], quat [4] (w, x, y, z), rumble, trigger / square / triangle / circle (ps_move.py:32-47)."""

    def __init__(self):
        self.values: Dict[str, Any] = dict(pos=np.zeros(3), quat=np.zeros(4), rumble=0, trigger=False, square=False,
                                           triangle=False, circle=False)

    def get(self, key: str):
        return self.values[key]

    def set(self, key: str, value: Any):
        self.values[key] = value


# (scene from, scene to, tracker from, tracker to) per axis: ps_move.py:96-114.  The tracker's x is the scene's x, its sphere
# radius the scene's y (forward / backward) and minus its y the scene's z (ps_move.py:148-153).
_RANGES = {
    MoveName.LEFT: ((0.2, -0.7, 375, 600), (0.9, 0.0, 12, 70), (0.01, 0.5, -400, -20)),
    MoveName.RIGHT: ((0.7, -0.2, 150, 375), (0.9, 0.0, 12, 70), (0.01, 0.5, -400, -20)),
}


def tracker_to_sim(move_name: MoveName, x: float, y: float, radius: float) -> np.ndarray:
    """Tracker reading (image x, image y, sphere radius) -> scene position: clamp into the tracker range, map linearly."""
    out = np.zeros(3)
    for dim, pos in enumerate((x, radius, -1.0 * y)):
        s0, s1, m0, m1 = _RANGES[move_name][dim]
        pos = min(max(pos, m0), m1)
        out[dim] = s0 + (pos - m0) / (m1 - m0) * (s1 - s0)
    return out


def move_orientation(move_quat) -> np.ndarray:
    """The controller's orientation as the loop wants it: first Euler angle kept, second dropped, THIRD slot fed with the
    second angle (ps_move.py:170-173)."""
    eul = quat2euler(move_quat)
    return euler2quat(eul[0], 0, eul[1])


def rumble_level(rumble: float) -> int:
    """Rumble motor level 0..130 from the (negative) gripper force the loop stores (ps_move.py:187-192)."""
    val = (-1.0 * rumble - 0.1) / (0.7 - 0.1) * 130
    return int(min(max(0, val), 130))


class PSMoveInterface:
    """move_states[MoveName] filled by `poll(readings)`; with no readings source the hardware collector needs the `psmove`
    module (a build of psmoveapi), which this package does not ship: it raises instead of pretending."""

    def __init__(self, multiprocess: bool = False, source=None):
        self.move_states = {name: MoveState() for name in MoveName}
        self.running = True
        self._source = source
        if source is None:
            try:
                import psmove  # noqa: F401
            except ImportError as e:
                raise RuntimeError("PSMoveInterface: no readings source was given and the psmove module is not installed") from e
            raise NotImplementedError("hardware collection needs psmoveapi's tracker; pass `source` (an iterator of readings)")

    def poll(self):
        """One pass of the collector over the injected source: {MoveName: dict(trigger_value, tracking, x, y, radius, buttons, quat)}."""
        reading = next(self._source)
        for name, r in reading.items():
            st = self.move_states[name]
            st.set("trigger", r.get("trigger_value", 0) > 10)
            st.set("triangle", bool(r.get("triangle", False)))
            st.set("circle", bool(r.get("circle", False)))
            st.set("quat", move_orientation(r["quat"]))
            if r.get("tracking", True):
                st.set("pos", tracker_to_sim(name, r["x"], r["y"], r["radius"]))
        return self.move_states

    def stop(self):
        self.running = False
