"""Operator layer of the PQ linear path: thin wrappers over the C ABI, registered as torch custom
ops with fake (meta) implementations so that torch.compile / graph capture can trace them.

Call-compatible with the reference's wrappers for its codebook kernels
(vllm/_custom_ops.py:347-371 fake impls, :646-658 wrappers):
    pq_gemm(input, codes, codebooks, scales, codebook_partition_sizes, bias)   <-> ops.aqlm_gemm
    pq_dequant(codes, codebooks, codebook_partition_sizes)                     <-> ops.aqlm_dequant
Unlike the reference's ops these accept every sub-vector length / codebook size / dtype of the
format (the reference's kernels take d == 8 with (K, M) in {(65536, 1), (256, 2)} and fp16 only,
csrc/quantization/aqlm/gemm_kernels.cu:519-542) and in_features is G * d rather than G * 8.

Tensors must live on a ROCm device; anything else raises.
"""
from typing import List, Optional

import torch

from . import _lib

_DTYPE = {torch.float16: _lib.PQ_DTYPE_FP16, torch.bfloat16: _lib.PQ_DTYPE_BF16}
_CODE_BYTES = {torch.int8: 1, torch.int16: 2}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device) -> int:
    """The current stream of `device` as the integer the C ABI takes (a call is on the host path of every launch:
    the raw query is several times cheaper than building a torch.cuda.Stream object)."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


class _on_device:
    """`with _on_device(dev):` -- torch.cuda.device(dev), entered only when dev is not the current device already."""
    __slots__ = ("guard",)

    def __init__(self, dev):
        self.guard = None if dev.index is None or dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            return self.guard.__exit__(*exc)
        return False


def _check_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                f"vllm_pq_amd ops run on MI355X only: got a tensor on '{t.device}'. There is no CPU fallback.")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"vllm_pq_amd: tensors on different devices ({dev} vs {t.device})")
    return dev


def _format(codes: torch.Tensor, codebooks: torch.Tensor, partition_sizes):
    if codes.dim() != 3:
        raise ValueError(f"codes must be [out_features, in_groups, num_codebooks], got {tuple(codes.shape)}")
    if codebooks.dim() != 4 or codebooks.shape[2] != 1:
        raise ValueError(f"codebooks must be [P*M, K, 1, d] (out_group_size 1), got {tuple(codebooks.shape)}")
    if codes.dtype not in _CODE_BYTES:
        raise ValueError(f"codes dtype {codes.dtype}: int8 and int16 are supported")
    if codebooks.dtype not in _DTYPE:
        raise ValueError(f"codebooks dtype {codebooks.dtype}: fp16 and bf16 are supported")
    N, G, M = codes.shape
    parts = [int(s) for s in partition_sizes] if len(partition_sizes) else [N]
    P = len(parts)
    if codebooks.shape[0] != P * M:
        raise ValueError(f"codebooks.shape[0]={codebooks.shape[0]} != partitions*num_codebooks={P}*{M}")
    K, d = codebooks.shape[1], codebooks.shape[3]
    return N, G, M, K, d, parts


def _raise(rc: int):
    msg = _lib.last_error()
    if rc in (1, 2):
        raise ValueError(msg)
    raise RuntimeError(msg)


# ---- workspace of the K-split launches ---------------------------------------------------------
# Zero-initialised buffers of ONE fixed size (the library's maximum, pq_gemm_workspace_max_bytes()), so that a
# buffer is never grown and thereby replaced: launches captured into a HIP graph keep pointing to it.  One buffer
# per (device, stream): launches on one stream run one after the other and each leaves the counters at the head
# zero again, launches on different streams may overlap and must not share.  A stream's buffer is created at
# weight-load time (ensure_workspace) or by the stream's first eager call.  Launches issued under graph capture use
# the device's one capture buffer, created at load time only (nothing is allocated under capture; without it they
# run unsplit): replays of different captured graphs must not overlap on one device.
_WORKSPACE = {}


def workspace_bytes(T: int, N: int, G: int, M: int, K: int, d: int, code_bytes: int, dtype: torch.dtype) -> int:
    """Bytes of scratch with which pq_gemm may split K over several blocks for this problem (0: never)."""
    return int(_lib.load().pq_gemm_workspace_bytes(T, N, G, M, K, d, code_bytes, _DTYPE[dtype]))


def _new_workspace(idx: int) -> torch.Tensor:
    return torch.zeros(int(_lib.load().pq_gemm_workspace_max_bytes()), dtype=torch.uint8, device=torch.device("cuda", idx))


def ensure_workspace(device: torch.device, nbytes: Optional[int] = None) -> None:
    """Create the workspaces of `device` for the current stream and for graph capture, if they do not exist yet.
    Call at load time, outside graph capture.  `nbytes` (a pq_gemm_workspace_bytes value) only says whether any
    is needed: 0 does nothing; the buffers always have the fixed maximum size."""
    if nbytes is not None and nbytes <= 0:
        return
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    for key in ((idx, _stream_ptr(torch.device("cuda", idx))), (idx, "graph")):
        if key not in _WORKSPACE:
            _WORKSPACE[key] = _new_workspace(idx)


def _workspace(dev: torch.device) -> Optional[torch.Tensor]:
    """The buffer a launch on the current stream of `dev` may use (None: run unsplit)."""
    if not _WORKSPACE:
        return None
    idx = dev.index
    if torch.cuda.is_current_stream_capturing():
        return _WORKSPACE.get((idx, "graph"))
    if (idx, "graph") not in _WORKSPACE:      # the device never asked for workspaces (or dropped them)
        return None
    key = (idx, _stream_ptr(dev))
    ws = _WORKSPACE.get(key)
    if ws is None:
        ws = _WORKSPACE[key] = _new_workspace(idx)
    return ws


def drop_workspace(device: Optional[torch.device] = None) -> None:
    """Forget the workspaces (of one device): later launches run unsplit until ensure_workspace is called again.
    Not while a captured graph that used them may still be replayed."""
    if device is None:
        _WORKSPACE.clear()
    else:
        idx = torch.device(device).index
        for key in [k for k in _WORKSPACE if k[0] == idx]:
            del _WORKSPACE[key]


# ---- eager implementations ------------------------------------------------------------------
def _pq_dequant_impl(codes: torch.Tensor, codebooks: torch.Tensor, codebook_partition_sizes: List[int],
                     scales: Optional[torch.Tensor] = None) -> torch.Tensor:
    L = _lib.load()
    dev = _check_device(codes, codebooks, scales)
    N, G, M, K, d, parts = _format(codes, codebooks, codebook_partition_sizes)
    codes = codes.contiguous()
    codebooks = codebooks.contiguous()
    if scales is not None:
        if scales.numel() != N or scales.dtype != codebooks.dtype:
            raise ValueError("scales must hold one value of the codebooks' dtype per output row")
        scales = scales.contiguous()
    out = torch.empty((N, G * d), dtype=codebooks.dtype, device=dev)
    with _on_device(dev):
        rc = L.pq_dequant(codes.data_ptr(), codebooks.data_ptr(), scales.data_ptr() if scales is not None else None,
                          _lib.parts_array(parts), len(parts), N, G, M, K, d, _CODE_BYTES[codes.dtype],
                          _DTYPE[codebooks.dtype], out.data_ptr(), _stream_ptr(dev))
    if rc:
        _raise(rc)
    return out


def _pq_embedding_impl(input_ids: torch.Tensor, codes: torch.Tensor, codebooks: torch.Tensor,
                       scales: Optional[torch.Tensor], codebook_partition_sizes: List[int]) -> torch.Tensor:
    L = _lib.load()
    dev = _check_device(input_ids, codes, codebooks, scales)
    N, G, M, K, d, parts = _format(codes, codebooks, codebook_partition_sizes)
    if input_ids.dtype not in (torch.int64, torch.int32):
        raise ValueError(f"input_ids dtype {input_ids.dtype}: int64 (or int32) indices expected")
    if scales is not None:
        if scales.numel() != N or scales.dtype != codebooks.dtype:
            raise ValueError("scales must hold one value of the codebooks' dtype per output row")
        scales = scales.contiguous()
    ids = input_ids.reshape(-1).to(torch.int64).contiguous()
    T = ids.numel()
    out = torch.empty((T, G * d), dtype=codebooks.dtype, device=dev)
    if T and G:
        with _on_device(dev):
            rc = L.pq_embedding(ids.data_ptr(), T, codes.contiguous().data_ptr(), codebooks.contiguous().data_ptr(),
                                scales.data_ptr() if scales is not None else None, _lib.parts_array(parts), len(parts),
                                N, G, M, K, d, _CODE_BYTES[codes.dtype], _DTYPE[codebooks.dtype], out.data_ptr(), G * d,
                                _stream_ptr(dev))
        if rc:
            _raise(rc)
    return out.reshape(tuple(input_ids.shape) + (G * d,))


def _pq_gemm_impl(input: torch.Tensor, codes: torch.Tensor, codebooks: torch.Tensor, scales: torch.Tensor,
                  codebook_partition_sizes: List[int], bias: Optional[torch.Tensor],
                  packed_codes: Optional[torch.Tensor] = None) -> torch.Tensor:
    L = _lib.load()
    dev = _check_device(input, codes, codebooks, scales, bias, packed_codes)
    N, G, M, K, d, parts = _format(codes, codebooks, codebook_partition_sizes)
    if input.dtype != codebooks.dtype:
        raise ValueError(f"input dtype {input.dtype} != codebooks dtype {codebooks.dtype}")
    if input.shape[-1] != G * d:
        raise ValueError(f"input has {input.shape[-1]} features, the weight expects {G * d}")
    if scales.numel() != N or scales.dtype != input.dtype:
        raise ValueError("scales must hold one value of the activation dtype per output row")
    if bias is not None and (bias.numel() != N or bias.dtype != input.dtype):
        raise ValueError("bias must hold one value of the activation dtype per output row")
    x2 = input.reshape(-1, input.shape[-1])
    if x2.stride(-1) != 1 or (packed_codes is not None and (x2.data_ptr() % 16 or x2.stride(0) % 8)):
        x2 = x2.contiguous()
    T = x2.shape[0]
    if packed_codes is None:
        codes = codes.contiguous()
    codebooks = codebooks.contiguous()
    scales = scales.contiguous()
    if bias is not None:
        bias = bias.contiguous()
    out = torch.empty((T, N), dtype=input.dtype, device=dev)
    if T and N:
        with _on_device(dev):
            ws = _workspace(dev)
            fn = L.pq_gemm if packed_codes is None else L.pq_gemm_packed
            cptr = codes.data_ptr() if packed_codes is None else packed_codes.data_ptr()
            rc = fn(x2.data_ptr(), T, x2.stride(0) if T > 1 else G * d, cptr,
                           codebooks.data_ptr(), scales.data_ptr(), bias.data_ptr() if bias is not None else None,
                           _lib.parts_array(parts), len(parts), N, G, M, K, d, _CODE_BYTES[codes.dtype],
                           _DTYPE[input.dtype], out.data_ptr(), N, ws.data_ptr() if ws is not None else None,
                           ws.numel() if ws is not None else 0, _stream_ptr(dev))
        if rc:
            _raise(rc)
    return out if input.dim() == 2 else out.reshape(input.shape[:-1] + (N,))


class PackedLinear:
    """A loaded layer's launch arguments, validated once (pq_gemm_packed checks shapes, dtypes and devices of five
    weight tensors on every call: several microseconds of host time, as long as a decode launch runs).  Holds the
    tensors it points into; `matches()` tells whether they are still the ones the layer holds."""
    __slots__ = ("packed", "codes", "codebooks", "scales", "ptrs", "N", "G", "M", "K", "d", "cb", "dtype", "tdtype",
                 "parts", "nparts", "device", "features")

    def __init__(self, packed_codes: torch.Tensor, codes: torch.Tensor, codebooks: torch.Tensor, scales: torch.Tensor,
                 codebook_partition_sizes: List[int]):
        dev = _check_device(packed_codes, codes, codebooks, scales)
        N, G, M, K, d, parts = _format(codes, codebooks, codebook_partition_sizes)
        if scales.numel() != N or scales.dtype != codebooks.dtype:
            raise ValueError("scales must hold one value of the activation dtype per output row")
        if not (codebooks.is_contiguous() and scales.is_contiguous() and packed_codes.is_contiguous()):
            raise ValueError("contiguous weight tensors expected")
        self.packed, self.codes, self.codebooks, self.scales = packed_codes, codes, codebooks, scales
        self.ptrs = (packed_codes.data_ptr(), codebooks.data_ptr(), scales.data_ptr(), packed_codes.numel(),
                     codebooks.numel(), scales.numel())
        self.N, self.G, self.M, self.K, self.d = N, G, M, K, d
        self.cb, self.dtype, self.tdtype = _CODE_BYTES[codes.dtype], _DTYPE[codebooks.dtype], codebooks.dtype
        self.parts, self.nparts = _lib.parts_array(parts), len(parts)
        self.device, self.features = dev, G * d

    def matches(self, packed_codes: torch.Tensor, codebooks: torch.Tensor, scales: torch.Tensor) -> bool:
        return self.ptrs == (packed_codes.data_ptr(), codebooks.data_ptr(), scales.data_ptr(), packed_codes.numel(),
                             codebooks.numel(), scales.numel())


def pq_gemm_packed_prepared(desc: PackedLinear, input: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pq_gemm_packed for a layer prepared with PackedLinear: the same launch, the per-call checks reduced to the
    activations and the bias.  Under torch.compile and for fake tensors it goes through the registered op."""
    if _via_op(input, bias):
        return torch.ops.pq_amd.gemm_packed(input, desc.packed, desc.codes, desc.codebooks, desc.scales,
                                            [desc.parts[i] for i in range(desc.nparts)], bias)
    dev, N = desc.device, desc.N
    if input.device != dev or input.dtype != desc.tdtype or input.shape[-1] != desc.features:
        if not input.is_cuda:
            raise RuntimeError(
                f"vllm_pq_amd ops run on MI355X only: got a tensor on '{input.device}'. There is no CPU fallback.")
        raise ValueError(f"input {tuple(input.shape)} {input.dtype} on {input.device} does not match the weight "
                         f"({desc.features} features, {desc.tdtype}, {dev})")
    bptr = None
    if bias is not None:
        if bias.device != dev or bias.numel() != N or bias.dtype != desc.tdtype:
            raise ValueError("bias must hold one value of the activation dtype per output row, on the weight's device")
        bptr = bias.contiguous().data_ptr()
    x2 = input if input.dim() == 2 else input.reshape(-1, desc.features)
    if x2.stride(-1) != 1 or x2.data_ptr() % 16 or x2.stride(0) % 8:
        x2 = x2.contiguous()
    T = x2.shape[0]
    out = torch.empty((T, N), dtype=desc.tdtype, device=dev)
    if T and N:
        with _on_device(dev):
            ws = _workspace(dev)
            rc = _lib.load().pq_gemm_packed(
                x2.data_ptr(), T, x2.stride(0) if T > 1 else desc.features, desc.ptrs[0], desc.ptrs[1], desc.ptrs[2], bptr,
                desc.parts, desc.nparts, N, desc.G, desc.M, desc.K, desc.d, desc.cb, desc.dtype, out.data_ptr(), N,
                ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream_ptr(dev))
        if rc:
            _raise(rc)
    return out if input.dim() == 2 else out.reshape(input.shape[:-1] + (N,))


# ---- torch custom ops (schema + fake impl), as the reference registers its ops ----------------
_HAVE_CUSTOM_OP = hasattr(torch, "library") and hasattr(torch.library, "custom_op")
_is_compiling = getattr(getattr(torch, "compiler", None), "is_compiling", None)


def _via_op(*tensors) -> bool:
    """Whether a call goes through its registered torch op: under torch.compile / tracing, and for anything that is
    not a plain CUDA tensor (fake and functional tensors).  Plain eager calls go straight to the implementation: the
    Python-level dispatch of a custom op costs about 7 us per call, more than a decode launch runs."""
    if not _HAVE_CUSTOM_OP:
        return False
    if _is_compiling is not None and _is_compiling():
        return True
    for t in tensors:
        if t is not None and (type(t) is not torch.Tensor and type(t) is not torch.nn.Parameter):
            return t.is_cuda
    return False

if _HAVE_CUSTOM_OP:

    @torch.library.custom_op("pq_amd::gemm", mutates_args=(), device_types="cuda")
    def _gemm_op(input: torch.Tensor, codes: torch.Tensor, codebooks: torch.Tensor, scales: torch.Tensor,
                 codebook_partition_sizes: List[int], bias: Optional[torch.Tensor]) -> torch.Tensor:
        return _pq_gemm_impl(input, codes, codebooks, scales, codebook_partition_sizes, bias)

    @_gemm_op.register_fake
    def _gemm_fake(input, codes, codebooks, scales, codebook_partition_sizes, bias):
        return input.new_empty(input.shape[:-1] + (codes.shape[0],))

    @torch.library.custom_op("pq_amd::gemm_packed", mutates_args=(), device_types="cuda")
    def _gemm_packed_op(input: torch.Tensor, packed_codes: torch.Tensor, codes: torch.Tensor,
                        codebooks: torch.Tensor, scales: torch.Tensor, codebook_partition_sizes: List[int],
                        bias: Optional[torch.Tensor]) -> torch.Tensor:
        return _pq_gemm_impl(input, codes, codebooks, scales, codebook_partition_sizes, bias, packed_codes)

    @_gemm_packed_op.register_fake
    def _gemm_packed_fake(input, packed_codes, codes, codebooks, scales, codebook_partition_sizes, bias):
        return input.new_empty(input.shape[:-1] + (codes.shape[0],))

    @torch.library.custom_op("pq_amd::dequant", mutates_args=(), device_types="cuda")
    def _dequant_op(codes: torch.Tensor, codebooks: torch.Tensor, codebook_partition_sizes: List[int],
                    scales: Optional[torch.Tensor]) -> torch.Tensor:
        return _pq_dequant_impl(codes, codebooks, codebook_partition_sizes, scales)

    @_dequant_op.register_fake
    def _dequant_fake(codes, codebooks, codebook_partition_sizes, scales):
        return codebooks.new_empty((codes.shape[0], codes.shape[1] * codebooks.shape[3]))


    @torch.library.custom_op("pq_amd::embedding", mutates_args=(), device_types="cuda")
    def _embedding_op(input_ids: torch.Tensor, codes: torch.Tensor, codebooks: torch.Tensor,
                      scales: Optional[torch.Tensor], codebook_partition_sizes: List[int]) -> torch.Tensor:
        return _pq_embedding_impl(input_ids, codes, codebooks, scales, codebook_partition_sizes)

    @_embedding_op.register_fake
    def _embedding_fake(input_ids, codes, codebooks, scales, codebook_partition_sizes):
        return codebooks.new_empty(tuple(input_ids.shape) + (codes.shape[1] * codebooks.shape[3],))


def pq_embedding(input_ids: torch.Tensor, codes: torch.Tensor, codebooks: torch.Tensor,
                 scales: Optional[torch.Tensor], codebook_partition_sizes: List[int]) -> torch.Tensor:
    """Rows `input_ids` of the dequantised weight, [..., in_features]: F.embedding(input_ids, pq_dequant(...)) bit
    for bit without building the table (UnquantizedEmbeddingMethod.embedding,
    vllm/model_executor/layers/vocab_parallel_embedding.py:45-47, for a PQ-compressed table)."""
    if _via_op(input_ids, codes):
        return torch.ops.pq_amd.embedding(input_ids, codes, codebooks, scales, list(codebook_partition_sizes))
    return _pq_embedding_impl(input_ids, codes, codebooks, scales, codebook_partition_sizes)


def pq_gemm(input: torch.Tensor, codes: torch.Tensor, codebooks: torch.Tensor, scales: torch.Tensor,
            codebook_partition_sizes: List[int], bias: Optional[torch.Tensor]) -> torch.Tensor:
    """y = input @ W^T (+ bias): W is decoded on the fly from codes/codebooks/scales.
    Same signature as ops.aqlm_gemm (vllm/_custom_ops.py:646-652)."""
    if _via_op(input, codes):
        return torch.ops.pq_amd.gemm(input, codes, codebooks, scales, list(codebook_partition_sizes), bias)
    return _pq_gemm_impl(input, codes, codebooks, scales, codebook_partition_sizes, bias)


def pq_gemm_packed(input: torch.Tensor, packed_codes: torch.Tensor, codes: torch.Tensor, codebooks: torch.Tensor,
                   scales: torch.Tensor, codebook_partition_sizes: List[int],
                   bias: Optional[torch.Tensor]) -> torch.Tensor:
    """pq_gemm reading the codes from `packed_codes` (pq_repack_codes); `codes` only supplies the
    shape and dtype of the reference-layout tensor."""
    if _via_op(input, packed_codes):
        return torch.ops.pq_amd.gemm_packed(input, packed_codes, codes, codebooks, scales,
                                            list(codebook_partition_sizes), bias)
    return _pq_gemm_impl(input, codes, codebooks, scales, codebook_partition_sizes, bias, packed_codes)


def pq_repack_codes(codes: torch.Tensor, codebooks: torch.Tensor,
                    codebook_partition_sizes: List[int]) -> Optional[torch.Tensor]:
    """Load-time re-layout of `codes` into the order the MFMA kernels stream (one contiguous KiB per
    16-row fragment and k-chunk).  Returns a uint8 tensor, or None when the format has no packed
    layout (the caller keeps using pq_gemm on the reference layout)."""
    L = _lib.load()
    dev = _check_device(codes, codebooks)
    N, G, M, K, d, parts = _format(codes, codebooks, codebook_partition_sizes)
    pa = _lib.parts_array(parts)
    nbytes = L.pq_packed_codes_bytes(pa, len(parts), N, G, M, K, d, _CODE_BYTES[codes.dtype])
    if nbytes == 0:
        return None
    codes = codes.contiguous()
    packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with _on_device(dev):
        rc = L.pq_repack_codes(codes.data_ptr(), pa, len(parts), N, G, M, K, d, _CODE_BYTES[codes.dtype],
                               packed.data_ptr(), _stream_ptr(dev))
    if rc:
        _raise(rc)
    return packed


def fold_pairs(codes: torch.Tensor, codebooks: torch.Tensor):
    """A format with 16-entry codebooks restated with 256-entry ones, bit for bit: two neighbouring groups of
    `d` features become one group of `2 d` whose entry is the two original entries side by side,
        W[o, 2 d g' : 2 d (g' + 1)] = concat(cb[c[o, 2 g']], cb[c[o, 2 g' + 1]]) = cb2[c[o, 2 g'] + 16 c[o, 2 g' + 1]],
    so the layer runs on the kernels of the format with twice the sub-vector length (16 entries x d = 4 becomes
    256 entries x d = 8: the decode, mid-batch and hand-pipelined prefill kernels; half the code bytes).  Returns
    (codes2 [N, G / 2, 1] int8, codebooks2 [partitions, 256, 1, 2 d]) to hand to pq_repack_codes and the packed entry
    points in place of the originals, or None when the format cannot be folded (one codebook of 16 entries, one-byte
    codes, an even number of groups, d = 2 or 4).  Load-time work: a few tensor operations on the device."""
    if codes.dim() != 3 or codebooks.dim() != 4 or codes.shape[2] != 1 or codebooks.shape[2] != 1:
        return None
    N, G, _ = codes.shape
    K, d = codebooks.shape[1], codebooks.shape[3]
    if K != 16 or d not in (2, 4) or G % 2 or codes.dtype != torch.int8:
        return None
    idx = codes[:, :, 0].to(torch.int16) & 15                      # index = stored value mod K (aqlm.py:32-34)
    both = idx[:, 0::2] | (idx[:, 1::2] << 4)                      # 0 .. 255
    codes2 = both.to(torch.uint8).view(torch.int8).unsqueeze(-1).contiguous()
    e = torch.arange(256, device=codebooks.device)
    cb = codebooks[:, :, 0, :]                                     # [P, 16, d]
    codebooks2 = torch.cat((cb[:, e & 15, :], cb[:, e >> 4, :]), dim=-1).unsqueeze(2).contiguous()
    return codes2, codebooks2


def pq_dequant(codes: torch.Tensor, codebooks: torch.Tensor, codebook_partition_sizes: List[int],
               scales: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Dense [out_features, in_features] weight.  With scales=None: same signature and result as
    ops.aqlm_dequant (vllm/_custom_ops.py:655-658); with scales it is dequantize_weight per partition
    (aqlm.py:37-74)."""
    if _via_op(codes, codebooks):
        return torch.ops.pq_amd.dequant(codes, codebooks, list(codebook_partition_sizes), scales)
    return _pq_dequant_impl(codes, codebooks, codebook_partition_sizes, scales)


# ---- gate_up_proj with its activation ------------------------------------------------------------
def _pq_gemm_packed_silu_mul_impl(input: torch.Tensor, packed_codes: torch.Tensor, codes: torch.Tensor,
                                  codebooks: torch.Tensor, scales: torch.Tensor,
                                  bias: Optional[torch.Tensor]) -> torch.Tensor:
    L = _lib.load()
    dev = _check_device(input, packed_codes, codes, codebooks, scales, bias)
    N, G, M, K, d, parts = _format(codes, codebooks, [codes.shape[0] // 2, codes.shape[0] - codes.shape[0] // 2])
    I = N // 2
    if N != 2 * I or M != 1:
        raise ValueError("pq_gemm_packed_silu_mul: two equal output partitions (gate, up) and one codebook expected")
    if input.dtype != codebooks.dtype or input.shape[-1] != G * d:
        raise ValueError(f"input {tuple(input.shape)} {input.dtype} does not match the weight ({G * d} features, "
                         f"{codebooks.dtype})")
    if scales.numel() != N or scales.dtype != input.dtype:
        raise ValueError("scales must hold one value of the activation dtype per output row")
    if bias is not None and (bias.numel() != N or bias.dtype != input.dtype):
        raise ValueError("bias must hold one value of the activation dtype per output row")
    x2 = input.reshape(-1, input.shape[-1])
    if x2.stride(-1) != 1 or x2.data_ptr() % 16 or x2.stride(0) % 8:
        x2 = x2.contiguous()
    T = x2.shape[0]
    out = torch.empty((T, I), dtype=input.dtype, device=dev)
    if T and I:
        cb = _CODE_BYTES[codes.dtype]
        fused = L.pq_gemm_silu_mul_is_fused(T, I, G, K, d, cb, _DTYPE[input.dtype])
        tmp = None if fused else torch.empty((T, N), dtype=input.dtype, device=dev)
        with _on_device(dev):
            ws = _workspace(dev)
            rc = L.pq_gemm_packed_silu_mul(
                x2.data_ptr(), T, x2.stride(0) if T > 1 else G * d, packed_codes.data_ptr(),
                codebooks.contiguous().data_ptr(), scales.contiguous().data_ptr(),
                bias.contiguous().data_ptr() if bias is not None else None, I, G, K, d, cb, _DTYPE[input.dtype],
                out.data_ptr(), I, tmp.data_ptr() if tmp is not None else None,
                ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream_ptr(dev))
        if rc:
            _raise(rc)
    return out.reshape(input.shape[:-1] + (I,))


if _HAVE_CUSTOM_OP:

    @torch.library.custom_op("pq_amd::gemm_packed_silu_mul", mutates_args=(), device_types="cuda")
    def _gemm_packed_silu_mul_op(input: torch.Tensor, packed_codes: torch.Tensor, codes: torch.Tensor,
                                 codebooks: torch.Tensor, scales: torch.Tensor,
                                 bias: Optional[torch.Tensor]) -> torch.Tensor:
        return _pq_gemm_packed_silu_mul_impl(input, packed_codes, codes, codebooks, scales, bias)

    @_gemm_packed_silu_mul_op.register_fake
    def _gemm_packed_silu_mul_fake(input, packed_codes, codes, codebooks, scales, bias):
        return input.new_empty(input.shape[:-1] + (codes.shape[0] // 2,))


def pq_gemm_packed_silu_mul(input: torch.Tensor, packed_codes: torch.Tensor, codes: torch.Tensor,
                            codebooks: torch.Tensor, scales: torch.Tensor,
                            bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """silu(g) * u with [g | u] = pq_gemm_packed(input, ...) of a gate/up layer (two equal output partitions):
    bit-identical to pq_gemm_packed followed by silu_and_mul, in one launch for small batches."""
    if _via_op(input, packed_codes):
        return torch.ops.pq_amd.gemm_packed_silu_mul(input, packed_codes, codes, codebooks, scales, bias)
    return _pq_gemm_packed_silu_mul_impl(input, packed_codes, codes, codebooks, scales, bias)


# ---- a fused add + RMSNorm carried by the PQ linears either side of it (decode batches) ------------
def _norm_pair_args(codes, codebooks, partition_sizes):
    N, G, M, K, d, parts = _format(codes, codebooks, partition_sizes)
    return N, G, M, K, d, parts, _lib.parts_array(parts), _CODE_BYTES[codes.dtype]


def gemm_add_norm_supported(T: int, codes: torch.Tensor, codebooks: torch.Tensor, partition_sizes) -> bool:
    """Whether pq_gemm_packed_add_norm runs a batch of T tokens on this layer (the GEMM in front of the norm)."""
    N, G, M, K, d, parts, pa, cb = _norm_pair_args(codes, codebooks, partition_sizes)
    return bool(_lib.load().pq_gemm_packed_add_norm_supported(T, pa, len(parts), N, G, M, K, d, cb, _DTYPE[codebooks.dtype]))


def gemm_normed_supported(T: int, codes: torch.Tensor, codebooks: torch.Tensor, partition_sizes) -> bool:
    """Whether pq_gemm_packed_normed runs a batch of T tokens on this layer (the GEMM behind the norm)."""
    N, G, M, K, d, parts, pa, cb = _norm_pair_args(codes, codebooks, partition_sizes)
    return bool(_lib.load().pq_gemm_packed_normed_supported(T, pa, len(parts), N, G, M, K, d, cb, _DTYPE[codebooks.dtype]))


def _rows16(t: torch.Tensor, cols: int, what: str) -> torch.Tensor:
    # a VIEW of the caller's tensor (the residual is updated in place through it): leading dimensions that do not
    # collapse -- a slice along the sequence dimension of a 3-D tensor -- are refused rather than silently copied
    try:
        t2 = t.view(-1, t.shape[-1])
    except RuntimeError:
        raise ValueError(f"{what}: the leading dimensions must collapse into rows without a copy "
                         f"(shape {tuple(t.shape)}, strides {tuple(t.stride())})") from None
    if t2.shape[-1] != cols or t2.stride(-1) != 1 or t2.data_ptr() % 16 or t2.stride(0) % 8:
        raise ValueError(f"{what}: [tokens, {cols}] with unit column stride, 16-byte aligned rows expected")
    return t2


def pq_gemm_packed_add_norm(input: torch.Tensor, packed_codes: torch.Tensor, codes: torch.Tensor, codebooks: torch.Tensor,
                            scales: torch.Tensor, codebook_partition_sizes: List[int], bias: Optional[torch.Tensor],
                            residual: torch.Tensor, norm_weight: torch.Tensor):
    """The linear in front of a fused add + RMSNorm: r = pq_gemm_packed(input, ...) + residual replaces `residual`
    in place; returns (xw, ssq): r * norm_weight in the act dtype, and the per-fragment sums of r^2 [N / 16, Tp] that
    pq_gemm_packed_normed turns into 1 / rms.  Raises NotImplementedError-like ValueError when the library does not
    fuse the problem: ask gemm_add_norm_supported() first."""
    L = _lib.load()
    dev = _check_device(input, packed_codes, codes, codebooks, scales, bias, residual, norm_weight)
    N, G, M, K, d, parts, pa, cb = _norm_pair_args(codes, codebooks, codebook_partition_sizes)
    if input.dtype != codebooks.dtype or input.shape[-1] != G * d:
        raise ValueError(f"input {tuple(input.shape)} {input.dtype} does not match the weight ({G * d} features, {codebooks.dtype})")
    x2 = input.reshape(-1, input.shape[-1])
    if x2.stride(-1) != 1 or x2.data_ptr() % 16 or x2.stride(0) % 8:
        x2 = x2.contiguous()
    T = x2.shape[0]
    res2 = _rows16(residual, N, "residual")
    if res2.shape[0] != T or residual.dtype != input.dtype or norm_weight.numel() != N or norm_weight.dtype != input.dtype:
        raise ValueError("residual [tokens, out_features] and norm_weight [out_features] of the activation dtype expected")
    xw = torch.empty((T, N), dtype=input.dtype, device=dev)
    ssq = torch.empty((N // 16, (T + 15) // 16 * 16), dtype=torch.float32, device=dev)
    if T and N:
        with _on_device(dev):
            ws = _workspace(dev)
            rc = L.pq_gemm_packed_add_norm(
                x2.data_ptr(), T, x2.stride(0) if T > 1 else G * d, packed_codes.data_ptr(), codebooks.contiguous().data_ptr(),
                scales.contiguous().data_ptr(), bias.contiguous().data_ptr() if bias is not None else None, pa, len(parts),
                N, G, M, K, d, cb, _DTYPE[input.dtype], res2.data_ptr(), res2.stride(0) if T > 1 else N,
                norm_weight.contiguous().data_ptr(), xw.data_ptr(), N, ssq.data_ptr(),
                ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream_ptr(dev))
        if rc:
            _raise(rc)
    return xw.reshape(input.shape[:-1] + (N,)), ssq


def pq_gemm_packed_normed(xw: torch.Tensor, ssq: torch.Tensor, eps: float, packed_codes: torch.Tensor, codes: torch.Tensor,
                          codebooks: torch.Tensor, scales: torch.Tensor, codebook_partition_sizes: List[int],
                          bias: Optional[torch.Tensor]) -> torch.Tensor:
    """The linear behind a fused add + RMSNorm: pq_gemm_packed(xw, ...) with every token's sum multiplied by
    rsqrt(mean of r^2 + eps) before scale and bias; (xw, ssq) come from pq_gemm_packed_add_norm."""
    L = _lib.load()
    dev = _check_device(xw, ssq, packed_codes, codes, codebooks, scales, bias)
    N, G, M, K, d, parts, pa, cb = _norm_pair_args(codes, codebooks, codebook_partition_sizes)
    x2 = _rows16(xw, G * d, "xw")
    T = x2.shape[0]
    if xw.dtype != codebooks.dtype or ssq.dtype != torch.float32 or tuple(ssq.shape) != (G * d // 16, (T + 15) // 16 * 16) \
            or not ssq.is_contiguous():
        raise ValueError("xw of the weight's dtype and ssq [in_features / 16, tokens rounded up to 16] fp32 expected")
    out = torch.empty((T, N), dtype=xw.dtype, device=dev)
    if T and N:
        with _on_device(dev):
            ws = _workspace(dev)
            rc = L.pq_gemm_packed_normed(
                x2.data_ptr(), T, x2.stride(0) if T > 1 else G * d, packed_codes.data_ptr(), codebooks.contiguous().data_ptr(),
                scales.contiguous().data_ptr(), bias.contiguous().data_ptr() if bias is not None else None, pa, len(parts),
                N, G, M, K, d, cb, _DTYPE[xw.dtype], ssq.data_ptr(), float(eps), out.data_ptr(), N,
                ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream_ptr(dev))
        if rc:
            _raise(rc)
    return out.reshape(xw.shape[:-1] + (N,))


def gemm_normed_silu_mul_supported(T: int, codes: torch.Tensor, codebooks: torch.Tensor) -> bool:
    """Whether pq_gemm_packed_normed_silu_mul runs a batch of T tokens on this gate/up layer."""
    N, G, M, K, d, parts, pa, cb = _norm_pair_args(codes, codebooks, [codes.shape[0] // 2, codes.shape[0] - codes.shape[0] // 2])
    if M != 1 or N % 2:
        return False
    return bool(_lib.load().pq_gemm_packed_normed_silu_mul_supported(T, N // 2, G, K, d, cb, _DTYPE[codebooks.dtype]))


def pq_gemm_packed_normed_silu_mul(xw: torch.Tensor, ssq: torch.Tensor, eps: float, packed_codes: torch.Tensor,
                                   codes: torch.Tensor, codebooks: torch.Tensor, scales: torch.Tensor,
                                   bias: Optional[torch.Tensor]) -> torch.Tensor:
    """pq_gemm_packed_normed of a gate/up layer (two equal output partitions) with SiLU-and-mul in the same launch:
    bit-identical to pq_gemm_packed_normed followed by silu_and_mul.  Ask gemm_normed_silu_mul_supported() first."""
    L = _lib.load()
    dev = _check_device(xw, ssq, packed_codes, codes, codebooks, scales, bias)
    N, G, M, K, d, parts, pa, cb = _norm_pair_args(codes, codebooks, [codes.shape[0] // 2, codes.shape[0] - codes.shape[0] // 2])
    I = N // 2
    if N != 2 * I or M != 1:
        raise ValueError("pq_gemm_packed_normed_silu_mul: two equal output partitions (gate, up) and one codebook expected")
    x2 = _rows16(xw, G * d, "xw")
    T = x2.shape[0]
    if xw.dtype != codebooks.dtype or ssq.dtype != torch.float32 or tuple(ssq.shape) != (G * d // 16, (T + 15) // 16 * 16) \
            or not ssq.is_contiguous():
        raise ValueError("xw of the weight's dtype and ssq [in_features / 16, tokens rounded up to 16] fp32 expected")
    if scales.numel() != N or scales.dtype != xw.dtype or (bias is not None and (bias.numel() != N or bias.dtype != xw.dtype)):
        raise ValueError("scales (and bias) must hold one value of the activation dtype per output row")
    out = torch.empty((T, I), dtype=xw.dtype, device=dev)
    if T and I:
        with _on_device(dev):
            rc = L.pq_gemm_packed_normed_silu_mul(
                x2.data_ptr(), T, x2.stride(0) if T > 1 else G * d, packed_codes.data_ptr(), codebooks.contiguous().data_ptr(),
                scales.contiguous().data_ptr(), bias.contiguous().data_ptr() if bias is not None else None, I, G, K, d, cb,
                _DTYPE[xw.dtype], ssq.data_ptr(), float(eps), out.data_ptr(), I, _stream_ptr(dev))
        if rc:
            _raise(rc)
    return out.reshape(xw.shape[:-1] + (I,))


# ---- the element-wise neighbours of the linears -------------------------------------------------
# Same names, argument order and in-place conventions as the reference's wrappers
# (vllm/_custom_ops.py:151-158 rms_norm / fused_add_rms_norm; torch.ops._C.silu_and_mul as called from
# vllm/model_executor/layers/activation.py:78-83).
def _rows(t: torch.Tensor, what: str):
    """(2-D view sharing t's memory, row stride); raises when t cannot be addressed as strided rows."""
    if t.dim() < 1 or t.stride(-1) != 1:
        raise ValueError(f"{what}: the last dimension must be contiguous")
    t2 = t.reshape(-1, t.shape[-1]) if t.dim() != 2 else t
    if t2.data_ptr() != t.data_ptr():
        raise ValueError(f"{what}: leading dimensions cannot be flattened without a copy")
    return t2, (t2.stride(0) if t2.shape[0] > 1 else t2.shape[1])


def _silu_and_mul_impl(out: torch.Tensor, input: torch.Tensor) -> None:
    L = _lib.load()
    dev = _check_device(out, input)
    if input.dtype not in _DTYPE or out.dtype != input.dtype:
        raise ValueError(f"silu_and_mul: fp16 / bf16 tensors of one dtype expected, got {input.dtype} -> {out.dtype}")
    I = input.shape[-1] // 2
    if input.shape[-1] != 2 * I or out.shape != input.shape[:-1] + (I,):
        raise ValueError(f"silu_and_mul: input {tuple(input.shape)} does not halve into out {tuple(out.shape)}")
    x2, xs = _rows(input, "silu_and_mul input")
    o2, os_ = _rows(out, "silu_and_mul out")
    with _on_device(dev):
        rc = L.pq_silu_and_mul(x2.data_ptr(), max(xs, 2 * I), x2.shape[0], I, _DTYPE[input.dtype], o2.data_ptr(),
                               max(os_, I), _stream_ptr(dev))
    if rc:
        _raise(rc)


def _rms_norm_impl(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor, epsilon: float) -> None:
    L = _lib.load()
    dev = _check_device(out, input, weight)
    H = input.shape[-1]
    if input.dtype not in _DTYPE or out.dtype != input.dtype or weight.dtype != input.dtype:
        raise ValueError("rms_norm: input, out and weight must share one of fp16 / bf16")
    if out.shape != input.shape or weight.shape != (H,):
        raise ValueError(f"rms_norm: shapes {tuple(input.shape)} -> {tuple(out.shape)}, weight {tuple(weight.shape)}")
    x2, xs = _rows(input, "rms_norm input")
    o2, os_ = _rows(out, "rms_norm out")
    with _on_device(dev):
        rc = L.pq_rms_norm(x2.data_ptr(), xs, weight.contiguous().data_ptr(), float(epsilon), x2.shape[0], H,
                           _DTYPE[input.dtype], o2.data_ptr(), os_, _stream_ptr(dev))
    if rc:
        _raise(rc)


def _fused_add_rms_norm_impl(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                             epsilon: float) -> None:
    L = _lib.load()
    dev = _check_device(input, residual, weight)
    H = input.shape[-1]
    if input.dtype not in _DTYPE or residual.dtype != input.dtype or weight.dtype != input.dtype:
        raise ValueError("fused_add_rms_norm: input, residual and weight must share one of fp16 / bf16")
    if residual.shape != input.shape or weight.shape != (H,):
        raise ValueError(f"fused_add_rms_norm: shapes {tuple(input.shape)} / {tuple(residual.shape)}, "
                         f"weight {tuple(weight.shape)}")
    x2, xs = _rows(input, "fused_add_rms_norm input")
    r2, rs = _rows(residual, "fused_add_rms_norm residual")
    with _on_device(dev):
        rc = L.pq_fused_add_rms_norm(x2.data_ptr(), xs, r2.data_ptr(), rs, weight.contiguous().data_ptr(),
                                     float(epsilon), x2.shape[0], H, _DTYPE[input.dtype], _stream_ptr(dev))
    if rc:
        _raise(rc)


if _HAVE_CUSTOM_OP:

    @torch.library.custom_op("pq_amd::silu_and_mul", mutates_args=("out",), device_types="cuda")
    def _silu_and_mul_op(out: torch.Tensor, input: torch.Tensor) -> None:
        _silu_and_mul_impl(out, input)

    @torch.library.custom_op("pq_amd::rms_norm", mutates_args=("out",), device_types="cuda")
    def _rms_norm_op(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor, epsilon: float) -> None:
        _rms_norm_impl(out, input, weight, epsilon)

    @torch.library.custom_op("pq_amd::fused_add_rms_norm", mutates_args=("input", "residual"), device_types="cuda")
    def _fused_add_rms_norm_op(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                               epsilon: float) -> None:
        _fused_add_rms_norm_impl(input, residual, weight, epsilon)


def silu_and_mul(out: torch.Tensor, input: torch.Tensor) -> None:
    """out[..., j] = silu(input[..., j]) * input[..., I + j]; `out` is written in place."""
    if _via_op(input, out):
        return torch.ops.pq_amd.silu_and_mul(out, input)
    _silu_and_mul_impl(out, input)


def rms_norm(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor, epsilon: float) -> None:
    """out = RMS-normalised input times weight (vllm/_custom_ops.py:151-153)."""
    if _via_op(input, out):
        return torch.ops.pq_amd.rms_norm(out, input, weight, float(epsilon))
    _rms_norm_impl(out, input, weight, epsilon)


def fused_add_rms_norm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, epsilon: float) -> None:
    """residual += input; input = RMS-norm(residual) * weight, both in place (vllm/_custom_ops.py:156-158)."""
    if _via_op(input, residual):
        return torch.ops.pq_amd.fused_add_rms_norm(input, residual, weight, float(epsilon))
    _fused_add_rms_norm_impl(input, residual, weight, epsilon)


def gemm_kernel_name(T: int, N: int, G: int, M: int, K: int, d: int, code_bytes: int, dtype: torch.dtype,
                     partition_sizes, packed: bool = False) -> str:
    """Name of the kernel pq_gemm selects for a problem (no GPU needed)."""
    import ctypes
    L = _lib.load()
    buf = ctypes.create_string_buffer(64)
    parts = [int(s) for s in partition_sizes] if len(partition_sizes) else [N]
    rc = L.pq_gemm_kernel_name(T, N, G, M, K, d, code_bytes, _DTYPE[dtype], _lib.parts_array(parts), len(parts),
                               int(packed), buf, 64)
    if rc:
        _raise(rc)
    return buf.value.decode()
