"""MI355X-native product-quantised linear layer for vLLM's quantization plugin surface.

Importing the package registers the "pq" quantization method (with vLLM's registry when vLLM is
importable, with the stand-alone mirror otherwise).  All compute goes through libpq_amd.so
(include/pq_amd.h); there is no CPU or PyTorch fallback.
"""
from . import ops  # noqa: F401
from .compat import HAVE_VLLM  # noqa: F401
from .pq import (PQ_METHOD_NAME, PQConfig, PQEmbeddingMethod, PQLinearMethod, get_int_dtype, register,  # noqa: F401
                 unpack_int_data)

register()

__version__ = "0.1.0"
