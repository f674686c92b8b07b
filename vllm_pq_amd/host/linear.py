"""Callers of the quantization method: the tensor-parallel linear layers (stand-alone mirror of
vllm/model_executor/layers/linear.py).  Same class names, constructor arguments, forward contract
and attribute-driven ("v1") weight loaders as the reference:

    LinearMethodBase            linear.py:87-118
    LinearBase                  linear.py:145-189
    ReplicatedLinear            linear.py:192-282
    ColumnParallelLinear        linear.py:285-442   (loader :370-408, forward :418-434)
    MergedColumnParallelLinear  linear.py:445-640   (loader :494-640, AQLM `is_metadata` :619-624)
    QKVParallelLinear           linear.py:700-1058  (sizes :766-782, loader :873-1058, metadata :1038-1044)
    RowParallelLinear           linear.py:1061-1229 (loader :1144-1180, forward :1193-1221)

Only the loader attributes a codebook method uses are understood (output_dim, input_dim,
packed_dim, pack_factor, is_metadata); GGUF / Marlin / bitsandbytes special cases are not part of
this path.  The layers never compute: they hand x to quant_method.apply and own bias policy and
collectives.
"""
from abc import abstractmethod
from typing import List, Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from .base_config import QuantizationConfig, QuantizeMethodBase
from .parallel_state import (divide, get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                             split_tensor_along_last_dim, tensor_model_parallel_all_gather,
                             tensor_model_parallel_all_reduce)
from .utils import set_weight_attrs


class LinearMethodBase(QuantizeMethodBase):

    @abstractmethod
    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError


class UnquantizedLinearMethod(LinearMethodBase):

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        weight = Parameter(torch.empty(sum(output_partition_sizes), input_size_per_partition, dtype=params_dtype),
                           requires_grad=False)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0})
        layer.register_parameter("weight", weight)
        set_weight_attrs(weight, extra_weight_attrs)

    def apply(self, layer, x, bias=None):
        return F.linear(x, layer.weight, bias)


class LinearBase(torch.nn.Module):

    def __init__(self, input_size: int, output_size: int, skip_bias_add: bool = False,
                 params_dtype: Optional[torch.dtype] = None, quant_config: Optional[QuantizationConfig] = None,
                 prefix: str = "", *, return_bias: bool = True):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.skip_bias_add = skip_bias_add
        self.params_dtype = params_dtype if params_dtype is not None else torch.get_default_dtype()
        self.prefix = prefix
        self.return_bias = return_bias
        if quant_config is None:
            self.quant_method: Optional[QuantizeMethodBase] = UnquantizedLinearMethod()
        else:
            self.quant_method = quant_config.get_quant_method(self, prefix=prefix)

    # ---- shared pieces of the subclasses ---------------------------------------------------
    def _make_bias(self, use_bias: bool, size: int, loader) -> None:
        if use_bias:
            self.bias = Parameter(torch.empty(size, dtype=self.params_dtype))
            set_weight_attrs(self.bias, {"output_dim": 0, "weight_loader": loader})
        else:
            self.register_parameter("bias", None)

    def _result(self, output: torch.Tensor):
        if not self.return_bias:
            return output
        return output, (self.bias if self.skip_bias_add else None)


def _copy_checked(dst: torch.Tensor, src: torch.Tensor) -> None:
    if src.dim() == 0:
        src = src.reshape(1)
    assert dst.shape == src.shape, f"Tried to load weights of size {tuple(src.shape)} to a parameter of size {tuple(dst.shape)}"
    dst.copy_(src)


class ReplicatedLinear(LinearBase):

    def __init__(self, input_size: int, output_size: int, bias: bool = True, skip_bias_add: bool = False,
                 params_dtype: Optional[torch.dtype] = None, quant_config: Optional[QuantizationConfig] = None,
                 prefix: str = "", *, return_bias: bool = True):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config, prefix,
                         return_bias=return_bias)
        assert self.quant_method is not None
        self.quant_method.create_weights(self, input_size, [output_size], input_size, output_size,
                                         self.params_dtype, weight_loader=self.weight_loader)
        self._make_bias(bias, output_size, self.weight_loader)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor):
        _copy_checked(param.data, loaded_weight)

    def forward(self, x: torch.Tensor):
        out = self.quant_method.apply(self, x, None if self.skip_bias_add else self.bias)
        return self._result(out)


class ColumnParallelLinear(LinearBase):
    """Y = X A + b with A split along its output dimension; no communication unless gather_output."""

    def __init__(self, input_size: int, output_size: int, bias: bool = True, gather_output: bool = False,
                 skip_bias_add: bool = False, params_dtype: Optional[torch.dtype] = None,
                 quant_config: Optional[QuantizationConfig] = None, output_sizes: Optional[List[int]] = None,
                 prefix: str = "", *, return_bias: bool = True):
        self.tp_size = get_tensor_model_parallel_world_size()
        self.input_size_per_partition = input_size
        self.output_size_per_partition = divide(output_size, self.tp_size)
        if hasattr(self, "output_sizes"):  # fused layers set it before calling up
            self.output_partition_sizes = [divide(s, self.tp_size) for s in self.output_sizes]
        else:
            self.output_partition_sizes = [self.output_size_per_partition]
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config, prefix,
                         return_bias=return_bias)
        self.gather_output = gather_output
        assert self.quant_method is not None
        self.quant_method.create_weights(layer=self, input_size_per_partition=self.input_size_per_partition,
                                         output_partition_sizes=self.output_partition_sizes,
                                         input_size=self.input_size, output_size=self.output_size,
                                         params_dtype=self.params_dtype, weight_loader=self.weight_loader)
        self._make_bias(bias, self.output_size_per_partition, self.weight_loader)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor):
        out_dim = getattr(param, "output_dim", None)
        data = param.data
        if out_dim is not None and not getattr(param, "is_sharded_weight", False):
            n = data.shape[out_dim]
            loaded_weight = loaded_weight.narrow(out_dim, get_tensor_model_parallel_rank() * n, n)
        _copy_checked(data, loaded_weight)

    def forward(self, input_: torch.Tensor):
        out = self.quant_method.apply(self, input_, None if self.skip_bias_add else self.bias)
        if self.gather_output:
            out = tensor_model_parallel_all_gather(out)
        return self._result(out)


def _packed(param, size: int, offset: int, dim: Optional[int]) -> Tuple[int, int]:
    """Row counts of a fused layer are in weight rows; a parameter packed along that dim stores
    pack_factor of them per element."""
    if dim is not None and getattr(param, "packed_dim", None) == dim:
        pf = param.pack_factor
        return size // pf, offset // pf
    return size, offset


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel matrices stacked along the output dimension (gate_up_proj); each is
    sharded on its own."""

    def __init__(self, input_size: int, output_sizes: List[int], bias: bool = True, gather_output: bool = False,
                 skip_bias_add: bool = False, params_dtype: Optional[torch.dtype] = None,
                 quant_config: Optional[QuantizationConfig] = None, prefix: str = "", *, return_bias: bool = True):
        self.output_sizes = list(output_sizes)
        tp = get_tensor_model_parallel_world_size()
        assert all(s % tp == 0 for s in output_sizes)
        super().__init__(input_size=input_size, output_size=sum(output_sizes), bias=bias,
                         gather_output=gather_output, skip_bias_add=skip_bias_add, params_dtype=params_dtype,
                         quant_config=quant_config, prefix=prefix, return_bias=return_bias)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor, loaded_shard_id: Optional[int] = None):
        out_dim = getattr(param, "output_dim", None)
        if loaded_shard_id is None:  # already fused on disk: cut it and load each piece
            if out_dim is None:
                _copy_checked(param.data, loaded_weight)
                return
            off = 0
            for i, sz in enumerate(self.output_sizes):
                s, o = _packed(param, sz, off, out_dim)
                self.weight_loader(param, loaded_weight.narrow(out_dim, o, s), i)
                off += sz
            return
        assert loaded_shard_id < len(self.output_sizes)
        tp_rank, tp = get_tensor_model_parallel_rank(), get_tensor_model_parallel_world_size()
        data = param.data
        if out_dim is not None:
            size, offset = _packed(param, self.output_sizes[loaded_shard_id] // tp,
                                   sum(self.output_sizes[:loaded_shard_id]) // tp, out_dim)
            data = data.narrow(out_dim, offset, size)
            if not getattr(param, "is_sharded_weight", False):
                loaded_weight = loaded_weight.narrow(out_dim, tp_rank * size, size)
        elif getattr(param, "is_metadata", False):
            # per-matrix blocks of equal size stacked along dim 0, replicated on every rank (codebooks)
            rows = loaded_weight.shape[0]
            data = data.narrow(0, loaded_shard_id * rows, rows)
        _copy_checked(data, loaded_weight)


class QKVParallelLinear(ColumnParallelLinear):
    """Fused q/k/v projection: query heads split over ranks, key/value heads split or replicated
    when there are fewer of them than ranks."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: Optional[int] = None, bias: bool = True, skip_bias_add: bool = False,
                 params_dtype: Optional[torch.dtype] = None, quant_config: Optional[QuantizationConfig] = None,
                 prefix: str = "", *, return_bias: bool = True):
        self.hidden_size = hidden_size
        self.head_size = head_size
        self.total_num_heads = total_num_heads
        self.total_num_kv_heads = total_num_heads if total_num_kv_heads is None else total_num_kv_heads
        tp = get_tensor_model_parallel_world_size()
        self.num_heads = divide(self.total_num_heads, tp)
        if tp >= self.total_num_kv_heads:
            self.num_kv_heads = 1
            self.num_kv_head_replicas = divide(tp, self.total_num_kv_heads)
        else:
            self.num_kv_heads = divide(self.total_num_kv_heads, tp)
            self.num_kv_head_replicas = 1
        self.output_sizes = [self.num_heads * head_size * tp, self.num_kv_heads * head_size * tp,
                             self.num_kv_heads * head_size * tp]
        super().__init__(input_size=hidden_size, output_size=sum(self.output_sizes), bias=bias, gather_output=False,
                         skip_bias_add=skip_bias_add, params_dtype=params_dtype, quant_config=quant_config,
                         prefix=prefix, return_bias=return_bias)

    def _slot(self, shard_id: str) -> Tuple[int, int]:
        """(offset, size) in weight rows of q / k / v inside this rank's fused output."""
        q = self.num_heads * self.head_size
        kv = self.num_kv_heads * self.head_size
        return {"q": (0, q), "k": (q, kv), "v": (q + kv, kv)}[shard_id]

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor, loaded_shard_id: Optional[str] = None):
        out_dim = getattr(param, "output_dim", None)
        if loaded_shard_id is None:  # fused on disk, full (unsharded) sizes
            if out_dim is None:
                _copy_checked(param.data, loaded_weight)
                return
            q = self.total_num_heads * self.head_size
            kv = self.total_num_kv_heads * self.head_size
            for sid, off, sz in (("q", 0, q), ("k", q, kv), ("v", q + kv, kv)):
                s, o = _packed(param, sz, off, out_dim)
                self.weight_loader(param, loaded_weight.narrow(out_dim, o, s), sid)
            return
        assert loaded_shard_id in ("q", "k", "v")
        tp_rank = get_tensor_model_parallel_rank()
        data = param.data
        if out_dim is not None:
            offset, size = self._slot(loaded_shard_id)
            size, offset = _packed(param, size, offset, out_dim)
            data = data.narrow(out_dim, offset, size)
            src = tp_rank if loaded_shard_id == "q" else tp_rank // self.num_kv_head_replicas
            if not getattr(param, "is_sharded_weight", False):
                loaded_weight = loaded_weight.narrow(out_dim, src * size, size)
        elif getattr(param, "is_metadata", False):
            rows = loaded_weight.shape[0]
            data = data.narrow(0, ("q", "k", "v").index(loaded_shard_id) * rows, rows)
        _copy_checked(data, loaded_weight)


class RowParallelLinear(LinearBase):
    """Y = X A + b with A split along its input dimension: each rank multiplies its slice of X and the
    partial results are summed by one all-reduce; the bias is added on rank 0 only."""

    def __init__(self, input_size: int, output_size: int, bias: bool = True, input_is_parallel: bool = True,
                 skip_bias_add: bool = False, params_dtype: Optional[torch.dtype] = None,
                 reduce_results: bool = True, quant_config: Optional[QuantizationConfig] = None, prefix: str = "",
                 *, return_bias: bool = True):
        self.tp_rank = get_tensor_model_parallel_rank()
        self.tp_size = get_tensor_model_parallel_world_size()
        self.input_size_per_partition = divide(input_size, self.tp_size)
        self.output_size_per_partition = output_size
        self.output_partition_sizes = [output_size]
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config, prefix,
                         return_bias=return_bias)
        self.input_is_parallel = input_is_parallel
        self.reduce_results = reduce_results
        assert self.quant_method is not None
        self.quant_method.create_weights(layer=self, input_size_per_partition=self.input_size_per_partition,
                                         output_partition_sizes=self.output_partition_sizes,
                                         input_size=self.input_size, output_size=self.output_size,
                                         params_dtype=self.params_dtype, weight_loader=self.weight_loader)
        if not reduce_results and (bias and not skip_bias_add):
            raise ValueError("When not reduce the results, adding bias to the results can lead to incorrect results")
        self._make_bias(bias, output_size, self.weight_loader)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor):
        in_dim = getattr(param, "input_dim", None)
        data = param.data
        if in_dim is not None and not getattr(param, "is_sharded_weight", False):
            n = data.shape[in_dim]
            loaded_weight = loaded_weight.narrow(in_dim, get_tensor_model_parallel_rank() * n, n)
        _copy_checked(data, loaded_weight)

    def forward(self, input_: torch.Tensor):
        if self.input_is_parallel:
            x = input_
        else:
            x = split_tensor_along_last_dim(input_, self.tp_size)[self.tp_rank].contiguous()
        bias = None if (self.tp_rank > 0 or self.skip_bias_add) else self.bias
        out = self.quant_method.apply(self, x, bias=bias)
        if self.reduce_results and self.tp_size > 1:
            out = tensor_model_parallel_all_reduce(out, in_place=True)     # apply()'s output is this call's own
        return self._result(out)

    def forward_add_norm(self, input_: torch.Tensor, residual: torch.Tensor, norm_weight: torch.Tensor, eps: float):
        """forward() followed by the decoder block's `RMSNorm.forward(out, residual)` (the pair a Llama layer runs
        behind o_proj and down_proj, vllm/model_executor/models/llama.py:272-284): `residual` becomes out + residual in
        place, the normalised tensor is returned.  With several ranks the sum over ranks, the add and the norm are one
        launch where the peer-buffer all-reduce takes the batch (decode sizes), the collective followed by the norm
        kernel otherwise -- the same values either way.  With one rank the add + norm rides on the GEMMs either side of
        it where the method fuses that (PQLinearMethod.apply_add_norm); this entry is for tp_size > 1 and plain layers."""
        from . import parallel_state
        from .. import ops
        if not (self.reduce_results and not self.skip_bias_add):
            raise ValueError("forward_add_norm: a layer that reduces its results and adds its bias itself")
        x = input_ if self.input_is_parallel else split_tensor_along_last_dim(input_, self.tp_size)[self.tp_rank].contiguous()
        out = self.quant_method.apply(self, x, bias=None if self.tp_rank > 0 else self.bias)
        if self.tp_size > 1:
            return parallel_state.tensor_model_parallel_all_reduce_add_norm(out, residual, norm_weight, eps)
        ops.fused_add_rms_norm(out, residual, norm_weight, eps)
        return out
