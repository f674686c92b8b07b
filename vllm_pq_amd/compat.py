"""Picks the interface classes the PQ method derives from: vLLM's own when vLLM is importable (the
drop-in case), this package's stand-alone mirror otherwise.  The vLLM branch is exercised against the reference's
real registry, base classes and parallel linear layers by tests/golden/make_golden_binding.py (re-run by
tests/test_binding_golden.py wherever the reference tree is present)."""
try:
    from vllm.model_executor.layers.linear import LinearBase, LinearMethodBase
    from vllm.model_executor.layers.quantization import (QUANTIZATION_METHODS, get_quantization_config,
                                                         register_quantization_config)
    from vllm.model_executor.layers.quantization.base_config import QuantizationConfig, QuantizeMethodBase
    from vllm.model_executor.layers.vocab_parallel_embedding import ParallelLMHead, VocabParallelEmbedding
    from vllm.model_executor.utils import set_weight_attrs
    HAVE_VLLM = True
except Exception:  # ImportError, or a half-installed vLLM
    from .host.base_config import (QUANTIZATION_METHODS, QuantizationConfig, QuantizeMethodBase,
                                   get_quantization_config, register_quantization_config)
    from .host.linear import LinearBase, LinearMethodBase
    from .host.utils import set_weight_attrs
    from .host.vocab_parallel_embedding import ParallelLMHead, VocabParallelEmbedding
    HAVE_VLLM = False

__all__ = ["HAVE_VLLM", "LinearBase", "LinearMethodBase", "ParallelLMHead", "VocabParallelEmbedding",
           "QUANTIZATION_METHODS", "QuantizationConfig",
           "QuantizeMethodBase", "get_quantization_config", "register_quantization_config", "set_weight_attrs"]
