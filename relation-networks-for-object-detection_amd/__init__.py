"""MI355X-native relation-network detection hot path (HIP kernels behind a C-ABI)."""
from . import lib  # noqa: F401

__all__ = ['lib']
