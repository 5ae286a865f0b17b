from irl_control_amd.transforms import quat2euler, euler2quat  # noqa: F401
