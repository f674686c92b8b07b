"""The element-wise layers either side of the PQ linears in a decoder block, with the reference's class names
and call signatures: `SiluAndMul` (vllm/model_executor/layers/activation.py:55-83) and `RMSNorm`
(vllm/model_executor/layers/layernorm.py:11-100).  `forward` is the reference's `forward_cuda` route: it
allocates the output and calls the library's kernels through `vllm_pq_amd.ops`; tensors that are not on a
ROCm device raise there (no PyTorch-native path is shipped).
"""
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import ops

__all__ = ["SiluAndMul", "RMSNorm"]


class SiluAndMul(nn.Module):
    """x -> silu(x[..., :d]) * x[..., d:], d = x.shape[-1] // 2."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        d = x.shape[-1] // 2
        out = torch.empty(x.shape[:-1] + (d,), dtype=x.dtype, device=x.device)
        ops.silu_and_mul(out, x)
        return out


class RMSNorm(nn.Module):
    """x -> w * x / sqrt(E[x^2] + eps); with `residual`: both are updated in place and returned, as the
    reference's fused kernel does (layernorm.py:86-92)."""

    def __init__(self, hidden_size: int, eps: float = 1e-6, var_hidden_size: Optional[int] = None,
                 has_weight: bool = True) -> None:
        super().__init__()
        if var_hidden_size is not None and var_hidden_size != hidden_size:
            # the reference routes this to its PyTorch-native path (layernorm.py:81-82); no kernel here takes it
            raise NotImplementedError("RMSNorm with a variance over part of the row is not supported")
        self.hidden_size = hidden_size
        self.variance_epsilon = eps
        self.has_weight = has_weight
        self.weight = torch.ones(hidden_size)
        if has_weight:
            self.weight = nn.Parameter(self.weight, requires_grad=False)

    def forward(self, x: torch.Tensor,
                residual: Optional[torch.Tensor] = None) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        if x.shape[-1] != self.hidden_size:
            raise ValueError(f"Expected hidden_size to be {self.hidden_size}, but found: {x.shape[-1]}")
        w = self.weight.data if self.has_weight else self.weight
        if w.device != x.device or w.dtype != x.dtype:
            w = w.to(device=x.device, dtype=x.dtype)
            if not self.has_weight:
                self.weight = w
        if residual is not None:
            ops.fused_add_rms_norm(x, residual, w, self.variance_epsilon)
            return x, residual
        out = torch.empty_like(x)
        ops.rms_norm(out, x, w, self.variance_epsilon)
        return out
