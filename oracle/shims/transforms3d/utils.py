from ._impl import normalized_vector  # noqa: F401
