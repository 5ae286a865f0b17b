from irl_control_amd.transforms import qconjugate  # noqa: F401
