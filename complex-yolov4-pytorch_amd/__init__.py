"""MI355X-native hot path for Complex-YOLOv4.  Host-side mirror of the reference's
model / geometry / post-processing interfaces over the C-ABI library ``libcyolo_hip.so``."""
__version__ = '0.1.0'

import os as _os

# The engine runs weight gradients on a side HIP stream and the data-parallel wrapper its all-reduce on another.  HIP maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and, with RCCL's streams in the process, all three land on one
# queue and serialise.  This only has an effect when set before the HIP runtime initialises (import this package, or set the
# variable, before the first device call).
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
