from irl_control_amd.transforms import compose  # noqa: F401
