from irl_control_amd.transforms import normalized_vector  # noqa: F401
