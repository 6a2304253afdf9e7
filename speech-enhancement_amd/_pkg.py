"""MI355X-native CMGAN / SCP-GAN hot path."""
from . import _lib  # noqa: F401
