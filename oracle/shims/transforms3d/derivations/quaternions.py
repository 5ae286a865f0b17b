from irl_control_amd.transforms import qmult  # noqa: F401
