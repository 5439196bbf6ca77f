"""btsbot_amd -- MI355X-native implementation of BTSbot's classifier forward/backward path.

Mirrors the public surface of the reference package for this path
(/root/reference/btsbot/__init__.py:9-46): the model classes of ``architectures`` and
``load_HF_model`` / ``download_HF_model``.  Importing this package never touches a GPU; the HIP
library is loaded the first time a model is constructed and there is no CPU fallback.
"""
__version__ = "0.1.0"

import os as _os

# ScoreStream (pipeline.py) overlaps consecutive batches on two or more HIP streams; the runtime multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4, the null stream included), and two streams that share a queue
# run one after the other.  Measured, 2 streams: 0.358 ms per batch at 4 queues, 0.341 at 8.  Read by the HIP runtime
# when it initialises (the first device call), so it only takes effect if that has not happened yet; an explicit
# setting of the user's wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import architectures
from . import from_HF
from . import to_HF
from . import data
from . import val
from .architectures import (
    MaxViT,
    ConvNeXt,
    mm_MaxViT,
    mm_ConvNeXt,
    mm_cnn,
    um_cnn,
    um_nn,
    frozen_fusion,
)
from .from_HF import download_HF_model, load_HF_model
from .synthetic import METADATA_COLS, synthetic_batch
from .pipeline import ScoreStream

__all__ = [
    "__version__", "architectures", "from_HF", "to_HF", "data", "val",
    "MaxViT", "ConvNeXt", "mm_MaxViT", "mm_ConvNeXt", "mm_cnn", "um_cnn", "um_nn", "frozen_fusion",
    "download_HF_model", "load_HF_model", "METADATA_COLS", "synthetic_batch", "ScoreStream",
]
