"""Twin of the reference's `lib/bbox` FFI surface: `bbox.bbox_overlaps_cython` (bbox.pyx:15-55) and the
`bbox_transform.bbox_overlaps` alias its callers import (bbox_transform.py:18-19)."""
from .bbox import bbox_overlaps_cython, bbox_overlaps  # noqa: F401
