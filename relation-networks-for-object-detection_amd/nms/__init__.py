"""Twin of the reference's `lib/nms` package (`nms.py`, `gpu_nms.pyx`, `cpu_nms.pyx`): same function names,
arguments and return values, every IoU / suppression decision made by `librelnet_hip.so`."""
from .nms import (gpu_nms, cpu_nms, nms, soft_nms, py_nms_wrapper, py_softnms_wrapper, cpu_nms_wrapper,  # noqa: F401
                  gpu_nms_wrapper)
