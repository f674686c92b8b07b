"""Plugin contract of a quantization method (mirror of
vllm/model_executor/layers/quantization/base_config.py:11-42 QuantizeMethodBase and :60-145
QuantizationConfig) and the method registry (quantization/__init__.py:8-73, :76-137)."""
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional, Type

import torch


class QuantizeMethodBase(ABC):
    """What a layer calls: create_weights once at construction, apply on every forward."""

    @abstractmethod
    def create_weights(self, layer: torch.nn.Module, *weight_args, **extra_weight_attrs):
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: torch.nn.Module, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError

    def embedding(self, layer: torch.nn.Module, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        return None


class QuantizationConfig(ABC):
    """Describes one quantization scheme and hands out the per-layer method object."""

    def __init__(self):
        super().__init__()
        self.packed_modules_mapping: Dict[str, List[str]] = {}

    @abstractmethod
    def get_name(self) -> str:
        raise NotImplementedError

    @abstractmethod
    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        raise NotImplementedError

    @classmethod
    @abstractmethod
    def get_min_capability(cls) -> int:
        raise NotImplementedError

    @staticmethod
    @abstractmethod
    def get_config_filenames() -> List[str]:
        raise NotImplementedError

    @classmethod
    @abstractmethod
    def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig":
        raise NotImplementedError

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        return None

    @staticmethod
    def get_from_keys(config: Dict[str, Any], keys: List[str]) -> Any:
        for k in keys:
            if k in config:
                return config[k]
        raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")

    @staticmethod
    def get_from_keys_or(config: Dict[str, Any], keys: List[str], default: Any) -> Any:
        try:
            return QuantizationConfig.get_from_keys(config, keys)
        except ValueError:
            return default

    @abstractmethod
    def get_quant_method(self, layer: torch.nn.Module, prefix: str) -> Optional[QuantizeMethodBase]:
        raise NotImplementedError

    def get_cache_scale(self, name: str) -> Optional[str]:
        return None


# ---- registry ---------------------------------------------------------------------------------
# The built-in names of the reference (quantization/__init__.py:8-34): a custom method may not
# reuse one of them.
QUANTIZATION_METHODS: List[str] = [
    "aqlm", "awq", "deepspeedfp", "tpu_int8", "fp8", "ptpc_fp8", "fbgemm_fp8", "modelopt", "marlin", "gguf",
    "gptq_marlin_24", "gptq_marlin", "awq_marlin", "gptq", "compressed-tensors", "bitsandbytes", "qqq", "hqq",
    "experts_int8", "neuron_quant", "ipex", "quark", "moe_wna16",
]
_BUILTIN = frozenset(QUANTIZATION_METHODS)
_CUSTOM: Dict[str, Type[QuantizationConfig]] = {}


def register_quantization_config(quantization: str):
    """Decorator: make `quantization` a valid method name backed by the decorated config class.
    Duplicate names and classes that are not QuantizationConfig subclasses raise ValueError
    (quantization/__init__.py:62-71)."""

    def wrap(cls):
        if quantization in QUANTIZATION_METHODS:
            raise ValueError(f"The quantization method `{quantization}` is already exists.")
        if not (isinstance(cls, type) and issubclass(cls, QuantizationConfig)):
            raise ValueError("The quantization config must be a subclass of `QuantizationConfig`.")
        _CUSTOM[quantization] = cls
        QUANTIZATION_METHODS.append(quantization)
        return cls

    return wrap


def get_quantization_config(quantization: str) -> Type[QuantizationConfig]:
    if quantization not in QUANTIZATION_METHODS:
        raise ValueError(f"Invalid quantization method: {quantization}")
    if quantization in _CUSTOM:
        return _CUSTOM[quantization]
    raise ValueError(f"quantization method `{quantization}` is a vLLM built-in; it is not part of this package")
