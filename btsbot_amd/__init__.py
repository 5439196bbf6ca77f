"""btsbot_amd -- MI355X-native implementation of BTSbot's classifier forward/backward path.

Mirrors the public surface of the reference package for this path
(/root/reference/btsbot/__init__.py:9-46): the model classes of ``architectures`` and
``load_HF_model`` / ``download_HF_model``.  Importing this package never touches a GPU; the HIP
library is loaded the first time a model is constructed and there is no CPU fallback.
"""
__version__ = "0.1.0"

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

__all__ = [
    "__version__", "architectures", "from_HF", "to_HF", "data", "val",
    "MaxViT", "ConvNeXt", "mm_MaxViT", "mm_ConvNeXt", "mm_cnn", "um_cnn", "um_nn", "frozen_fusion",
    "download_HF_model", "load_HF_model", "METADATA_COLS", "synthetic_batch",
]
