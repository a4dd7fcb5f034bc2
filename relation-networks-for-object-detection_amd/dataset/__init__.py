"""Data side of the path: COCO image database, image pre-processing, the loaders that feed
the Detector / Trainer, bbox evaluation, and the precomputed-proposal (alternate training / FPN) flow.
Host-side Python like the reference's `lib/dataset`, `lib/utils/image.py`, `core/loader.py`, `core/tester.py`;
tensors are handed to the HIP path as torch device tensors."""
from .imdb import IMDB  # noqa: F401
from .coco import coco  # noqa: F401
