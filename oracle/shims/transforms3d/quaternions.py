from ._impl import qconjugate, qmult, quat2mat  # noqa: F401
