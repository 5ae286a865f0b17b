"""MI355X-native batched operational-space controller with the irl_control API surface."""
from .version import version as __version__  # noqa: F401
