"""PyTorch-CPU restatement of the reference's CPU algorithm for this path, used ONLY as the timed
`cpu_baseline` of bench.py and as a second checker in tests.

TEST / MEASUREMENT INFRASTRUCTURE: never imported by the product package.

The reference runs plain PQ (and every format its CUDA kernels do not take) through
generic_dequantize_gemm (vllm/model_executor/layers/quantization/aqlm.py:94-127): for each output
partition, widen the codes to int64 and reduce them mod K (:32-34), gather-sum the codebook entries
with embedding_bag (:57-64), scale the rows (:70-72), reorder to [out, in] (:73-74) and call
F.linear (:90).  This module states the same steps with the same torch operators, so that timing it
on the GPU box's host cores measures what the reference's CPU path would cost there.
Pinned against the golden vectors in tests/test_oracle_golden.py (bit-exact: same operators).
"""
from typing import List, Optional

import torch
import torch.nn.functional as F


@torch.inference_mode()
def reconstruct_partition(codes: torch.Tensor, codebooks: torch.Tensor, scales: Optional[torch.Tensor]):
    """codes [n, G, M] (stored ints), codebooks [M, K, 1, d], scales [n,1,1,1] -> W [n, G*d]."""
    n, G, M = codes.shape
    K, d = codebooks.shape[1], codebooks.shape[3]
    nbits = K.bit_length() - 1
    idx = codes.to(torch.int64) % (1 << nbits)
    idx = idx.reshape(n * G, M) + torch.arange(0, M * K, K)
    w = F.embedding_bag(idx, codebooks.reshape(M * K, d), mode="sum").view(n, G, 1, d)
    if scales is not None:
        w = w * scales
    return w.transpose(1, 2).reshape(n, G * d)


@torch.inference_mode()
def pq_linear_cpu(x: torch.Tensor, codes: torch.Tensor, codebooks: torch.Tensor, scales: torch.Tensor,
                  partition_sizes: List[int], bias: Optional[torch.Tensor]) -> torch.Tensor:
    N = codes.shape[0]
    parts = list(partition_sizes) if len(partition_sizes) else [N]
    M = codebooks.shape[0] // len(parts)
    assert sum(parts) == N == scales.shape[0]
    out = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype)
    row = 0
    for p, size in enumerate(parts):
        w = reconstruct_partition(codes[row:row + size], codebooks[p * M:(p + 1) * M], scales[row:row + size])
        out[..., row:row + size] = F.linear(x, w, None if bias is None else bias[row:row + size])
        row += size
    return out
