"""Tensor-parallel group of the stand-alone host mirror: one process per GPU over
torch.distributed ("nccl" is RCCL on ROCm; "gloo" for CPU tests).

Mirrors what the linear layers need from vllm/distributed: get_tensor_model_parallel_rank /
_world_size, tensor_model_parallel_all_reduce / _all_gather (communication_op.py:11-19), divide and
split_tensor_along_last_dim (distributed/utils.py).  World size 1 bypasses every collective
(parallel_state.py:303).
"""
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

_TP_GROUP: Optional["dist.ProcessGroup"] = None
_TP_RANK = 0
_TP_SIZE = 1
_XGMI_AR = None          # XgmiAllReduce of the TP group, when enabled


def initialize_model_parallel(tensor_model_parallel_size: Optional[int] = None,
                              group: Optional["dist.ProcessGroup"] = None) -> None:
    """Use the whole default process group (or `group`) as the TP group."""
    global _TP_GROUP, _TP_RANK, _TP_SIZE
    if not dist.is_initialized():
        if tensor_model_parallel_size not in (None, 1):
            raise RuntimeError("torch.distributed is not initialised")
        _TP_GROUP, _TP_RANK, _TP_SIZE = None, 0, 1
        return
    g = group if group is not None else dist.group.WORLD
    size = dist.get_world_size(g)
    if tensor_model_parallel_size is not None and tensor_model_parallel_size != size:
        raise ValueError(f"tensor_model_parallel_size={tensor_model_parallel_size} != group size {size}")
    _TP_GROUP, _TP_RANK, _TP_SIZE = g, dist.get_rank(g), size


def enable_xgmi_all_reduce(max_bytes: Optional[int] = None) -> bool:
    """Collective over the TP group (2 to 8 GPUs of one node): from now on tensor_model_parallel_all_reduce sums
    small contiguous fp16 / bf16 / fp32 GPU tensors with the library's peer-buffer all-reduce (one launch, no
    ring) and everything else with RCCL, as the reference's communicator prefers its custom all-reduce by size
    (vllm/distributed/device_communicators/custom_all_reduce.py should_custom_ar).  False when not applicable."""
    global _XGMI_AR
    if _TP_SIZE < 2 or _TP_SIZE > 8 or not torch.cuda.is_available():
        return False
    from .xgmi_all_reduce import XgmiAllReduce, peers_reachable
    # one host, peer access between every pair of devices -- agreed on by all ranks; otherwise RCCL stays the path
    if not peers_reachable(_TP_GROUP):
        return False
    try:
        _XGMI_AR = XgmiAllReduce(_TP_GROUP, max_bytes)
    except RuntimeError:        # raised on every rank together (a buffer could not be allocated or mapped)
        _XGMI_AR = None
        return False
    return True


def disable_xgmi_all_reduce() -> None:
    global _XGMI_AR
    if _XGMI_AR is not None:
        _XGMI_AR.close()
        _XGMI_AR = None


def destroy_model_parallel() -> None:
    global _TP_GROUP, _TP_RANK, _TP_SIZE
    disable_xgmi_all_reduce()
    _TP_GROUP, _TP_RANK, _TP_SIZE = None, 0, 1


def get_tensor_model_parallel_rank() -> int:
    return _TP_RANK


def get_tensor_model_parallel_world_size() -> int:
    return _TP_SIZE


def tensor_model_parallel_all_reduce(input_: torch.Tensor, in_place: bool = False) -> torch.Tensor:
    """SUM over the TP group (RowParallelLinear.forward, linear.py:1212-1213).  Out of place by default, as the
    reference's op is for its callers; `in_place=True` reduces into `input_` itself and returns it -- for a caller
    that owns the tensor (the row-parallel layer: its GEMM output is fresh), which saves a copy per call."""
    if _TP_SIZE == 1:
        return input_
    if _XGMI_AR is not None and _XGMI_AR.should_use(input_):
        return _XGMI_AR.all_reduce(input_, input_ if in_place else None)
    out = input_ if in_place and input_.is_contiguous() else input_.clone(memory_format=torch.contiguous_format)
    if out.is_cuda and dist.get_backend(_TP_GROUP) == "gloo":
        # a rehearsal of the multi-rank path without RCCL (several ranks on one GPU): gloo's own handling of device
        # tensors was seen to hang between processes that share a device, so the sum goes through the host
        host = out.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=_TP_GROUP)
        out.copy_(host)
        return out
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=_TP_GROUP)
    return out


def tensor_model_parallel_all_reduce_add_norm(input_: torch.Tensor, residual: torch.Tensor, norm_weight: torch.Tensor,
                                             eps: float) -> torch.Tensor:
    """The sum of a row-parallel layer's partial outputs over the TP group followed by the decoder block's fused
    residual add + RMSNorm (RowParallelLinear.forward's all-reduce, linear.py:1212-1213, then
    RMSNorm.forward(x, residual), models/llama.py:272-284): `residual` becomes sum + residual in place, the
    normalised tensor is returned.  One launch of the peer-buffer all-reduce where it takes the batch (decode
    sizes); otherwise -- and with one rank -- the collective and the norm kernel, with the same values."""
    from .. import ops
    if _TP_SIZE > 1 and _XGMI_AR is not None and _XGMI_AR.add_norm_supported(input_, residual):
        return _XGMI_AR.all_reduce_add_norm(input_, residual, norm_weight, eps)
    x = tensor_model_parallel_all_reduce(input_, in_place=True)
    ops.fused_add_rms_norm(x, residual, norm_weight, eps)
    return x


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """Concatenate the ranks' tensors along `dim` (ColumnParallelLinear gather_output, linear.py:426-428)."""
    if _TP_SIZE == 1:
        return input_
    if dim < 0:
        dim += input_.dim()
    parts = [torch.empty_like(input_) for _ in range(_TP_SIZE)]
    dist.all_gather(parts, input_.contiguous(), group=_TP_GROUP)
    return torch.cat(parts, dim=dim)


def divide(numerator: int, denominator: int) -> int:
    if numerator % denominator != 0:
        raise AssertionError(f"{numerator} is not divisible by {denominator}")
    return numerator // denominator


def split_tensor_along_last_dim(tensor: torch.Tensor, num_partitions: int) -> Sequence[torch.Tensor]:
    size = divide(tensor.shape[-1], num_partitions)
    return torch.split(tensor, size, dim=-1)
