from ._impl import compose  # noqa: F401
