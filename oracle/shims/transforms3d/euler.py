from ._impl import euler2mat, euler2quat, mat2euler, quat2euler, quat2mat  # noqa: F401
