"""PQConfig / PQLinearMethod: the product-quantised linear method behind vLLM's
QuantizationConfig / LinearMethodBase plugin surface.

The tensor format is the reference's codebook format (AQLM, num_codebooks == 1 is plain PQ with a
per-partition codebook and a per-output-row scale):

    W[o, g*d:(g+1)*d] = scales[o] * sum_m codebooks[p(o)*M + m, codes[o, g, m] mod K, 0, :]

and every class / method mirrors its counterpart in
vllm/model_executor/layers/quantization/aqlm.py:
    get_int_dtype       :20-29     unpack_int_data     :32-34
    PQConfig            <-> AQLMConfig :159-217
    PQLinearMethod      <-> AQLMLinearMethod :220-374 (create_weights :230-323, apply :325-374)
    PQEmbeddingMethod   the same parameters on a vocabulary table: apply() for the LM head, embedding() for the
                        input table (QuantizeMethodBase.embedding, base_config.py:30-35); opt-in, see PQConfig
Differences, all on purpose: bf16 as well as fp16 activations; every (K, d, M) goes to the fused HIP
kernels (the reference only accelerates d == 8 with (K, M) in {(256, 2), (65536, 1)} and falls back to
PyTorch otherwise, aqlm.py:344-374); there is no CPU path here.  A format whose codebook does not fit LDS
(65536 entries) has no MFMA kernel: up to 8 tokens it runs on the generic fused kernel, beyond that as the
library's dequantisation kernel followed by a dense GEMM, the reference's own route for such batches.
"""
from typing import Any, Dict, List, Optional

import torch
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from . import ops
from .compat import (LinearBase, LinearMethodBase, ParallelLMHead, QUANTIZATION_METHODS, QuantizationConfig,
                     VocabParallelEmbedding, register_quantization_config, set_weight_attrs)

PQ_METHOD_NAME = "pq"


def get_int_dtype(nbits: int) -> torch.dtype:
    """Narrowest signed storage type holding an nbits-wide code (one code per element)."""
    for limit, dt in ((8, torch.int8), (16, torch.int16), (32, torch.int32), (64, torch.int64)):
        if nbits <= limit:
            return dt
    raise ValueError(f"No dtype available for {nbits}-bit codebooks")


@torch.inference_mode()
def unpack_int_data(data: torch.Tensor, nbits: int) -> torch.Tensor:
    """Signed stored codes -> unsigned indices in [0, 2**nbits)."""
    return data.to(torch.int64) % (1 << nbits)


class PQConfig(QuantizationConfig):
    """Config of the PQ method.  Keys of the HF `quantization_config` dict are the reference's:
    in_group_size (sub-vector length d), nbits_per_codebook (log2 K), num_codebooks (M, 1 = plain PQ),
    out_group_size (must be 1)."""

    def __init__(self, in_group_size: int = 8, nbits_per_codebook: int = 8, num_codebooks: int = 1,
                 out_group_size: int = 1, lm_head: bool = False, embeddings: bool = False,
                 release_reference_codes: bool = False) -> None:
        super().__init__()
        # After the load-time re-layout the kernels read only the packed copy of a linear layer's codes; with this
        # key (same name in the HF quantization_config) the reference-layout tensor gives its storage back (shape and
        # dtype stay): half the code bytes of a model, at the price that pq_dequant on such a layer no longer works.
        self.release_reference_codes = bool(release_reference_codes)
        # The reference's codebook method quantises linear layers only (aqlm.py:213-217), and so does this one by
        # default.  `lm_head` / `embeddings` (keys of the same names in the HF quantization_config) say that the
        # checkpoint also holds codes/codebooks/scales for the output head / the input embedding table.
        self.lm_head_quantized = bool(lm_head)
        self.embeddings_quantized = bool(embeddings)
        self.in_group_size = int(in_group_size)
        self.nbits_per_codebook = int(nbits_per_codebook)
        self.num_codebooks = int(num_codebooks)
        self.out_group_size = int(out_group_size)
        # the reference asserts this (aqlm.py:178-179)
        assert self.out_group_size == 1, "out_group_size > 1 is not supported"
        if self.in_group_size < 1 or self.num_codebooks < 1 or not (1 <= self.nbits_per_codebook <= 16):
            raise ValueError(f"unsupported PQ format: {self!r}")
        self.pack_factor = self.in_group_size * self.out_group_size

    def __repr__(self) -> str:
        return (f"PQConfig(in_group_size={self.in_group_size}, nbits_per_codebook={self.nbits_per_codebook}, "
                f"num_codebooks={self.num_codebooks}, out_group_size={self.out_group_size})")

    @classmethod
    def get_name(cls) -> str:
        return PQ_METHOD_NAME

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        # gfx950 reports compute capability 9.5; the kernels use gfx950-only MFMA shapes.  Platforms
        # without a capability (CPU) skip this gate (vllm/config.py:3333-3335).
        return 95

    @staticmethod
    def get_config_filenames() -> List[str]:
        return []

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "PQConfig":
        d = cls.get_from_keys(config, ["in_group_size", "sub_dim"])
        if "nbits_per_codebook" in config:
            nbits = config["nbits_per_codebook"]
        else:
            k = cls.get_from_keys(config, ["nbits_per_codebook", "codebook_size"])
            if k < 2 or k & (k - 1):
                raise ValueError(f"codebook_size={k} is not a power of two")
            nbits = k.bit_length() - 1
        m = cls.get_from_keys_or(config, ["num_codebooks"], 1)
        og = cls.get_from_keys_or(config, ["out_group_size"], 1)
        return cls(d, nbits, m, og, lm_head=cls.get_from_keys_or(config, ["lm_head"], False),
                   embeddings=cls.get_from_keys_or(config, ["embeddings"], False),
                   release_reference_codes=cls.get_from_keys_or(config, ["release_reference_codes"], False))

    def get_quant_method(self, layer: torch.nn.Module, prefix: str) -> Optional["PQLinearMethod"]:
        if isinstance(layer, LinearBase):
            return PQLinearMethod(self)
        if isinstance(layer, ParallelLMHead):
            return PQEmbeddingMethod(self) if self.lm_head_quantized else None
        if isinstance(layer, VocabParallelEmbedding):
            return PQEmbeddingMethod(self) if self.embeddings_quantized else None
        return None


class PQLinearMethod(LinearMethodBase):
    """Linear method for PQ-compressed weights."""

    def __init__(self, quant_config: PQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        del input_size, output_size
        cfg = self.quant_config
        if params_dtype not in (torch.half, torch.bfloat16):
            raise ValueError("Only half and bfloat16 are currently supported by pq")
        if input_size_per_partition % cfg.in_group_size != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        n_out = sum(output_partition_sizes)
        if n_out % cfg.out_group_size != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")

        # [out, in_groups, M] signed codes; rows shard along dim 0 (column parallel), groups along
        # dim 1 (row parallel); packed_dim/pack_factor let the fused-layer loaders convert offsets.
        codes = Parameter(torch.empty(n_out, input_size_per_partition // cfg.pack_factor, cfg.num_codebooks,
                                      dtype=get_int_dtype(cfg.nbits_per_codebook)), requires_grad=False)
        set_weight_attrs(codes, {"input_dim": 1, "output_dim": 0, "packed_dim": 1, "pack_factor": cfg.pack_factor})

        # one block of M codebooks per logical matrix, replicated on every rank
        codebooks = Parameter(torch.empty(cfg.num_codebooks * len(output_partition_sizes),
                                          2 ** cfg.nbits_per_codebook, cfg.out_group_size, cfg.in_group_size,
                                          dtype=params_dtype), requires_grad=False)
        set_weight_attrs(codebooks, {"is_metadata": True, "output_partition_sizes": list(output_partition_sizes)})

        scales = Parameter(torch.empty((n_out // cfg.out_group_size, 1, 1, 1), dtype=params_dtype),
                           requires_grad=False)
        set_weight_attrs(scales, {"output_dim": 0, "packed_dim": 0, "pack_factor": cfg.out_group_size})

        for name, p in (("codes", codes), ("codebooks", codebooks), ("scales", scales)):
            layer.register_parameter(name, p)
            set_weight_attrs(p, extra_weight_attrs)

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        # Integer codes left uninitialised by a dummy load are valid indices as they are
        # (index = value mod K).
        for name in ("codes", "codebooks", "scales"):
            p = getattr(layer, name)
            if not p.data.is_contiguous():
                p.data = p.data.contiguous()
        # Re-lay the codes out in the order the MFMA kernels stream them (kept beside the reference
        # layout, which the checkpoint loaders and pq_dequant use).  On a CPU-resident layer (weights
        # not moved to the GPU yet) there is nothing to do: apply() needs the GPU anyway.
        layer.pq_packed_codes = None
        layer.pq_prepared = None
        layer.pq_dequant_route = False
        layer.pq_folded = None
        if layer.codes.is_cuda:
            parts = getattr(layer.codebooks, "output_partition_sizes", [])
            # 16-entry codebooks: neighbouring groups folded in pairs, the same weights as a 256-entry format of twice
            # the sub-vector length (ops.fold_pairs), which is what the packed entry points then see
            folded = None if isinstance(self, PQEmbeddingMethod) else ops.fold_pairs(layer.codes.data, layer.codebooks.data)
            if folded is not None:
                codes2, codebooks2 = folded
                packed2 = ops.pq_repack_codes(codes2, codebooks2, parts)
                if packed2 is not None:
                    # (of the folded codes only shape, dtype and device are read from here on)
                    meta = torch.zeros(1, dtype=codes2.dtype, device=codes2.device).expand(codes2.shape)
                    layer.pq_folded = (meta, codebooks2)
                    layer.pq_packed_codes = packed2
            if layer.pq_packed_codes is None:
                layer.pq_packed_codes = ops.pq_repack_codes(layer.codes.data, layer.codebooks.data, parts)
            # Formats no MFMA kernel takes (a codebook that does not fit LDS, such as AQLM's 1x16 with 65536
            # entries) run on the generic one-wave-per-row kernel: fine for a few tokens, far too slow for a
            # prefill.  Larger batches take the route the reference's own fast path takes there
            # (optimized_dequantize_gemm, aqlm.py:132-156): dequantise, then a dense GEMM.
            pc, pcb = self._packed_operands(layer)      # the format the packed entry points see
            N, G, M = pc.shape
            K, d = pcb.shape[1], pcb.shape[3]
            layer.pq_dequant_route = ops.gemm_kernel_name(
                64, N, G, M, K, d, layer.codes.element_size(), layer.codebooks.dtype, parts or [N],
                packed=layer.pq_packed_codes is not None) == "generic"
            if (layer.pq_packed_codes is not None and self.quant_config.release_reference_codes
                    and not isinstance(self, PQEmbeddingMethod)):
                # a one-element tensor expanded to the old shape: what the operator layer reads of it (shape, dtype,
                # device) stays, the bytes go; the embedding method keeps its codes (pq_embedding reads them)
                layer.codes.data = torch.zeros(1, dtype=layer.codes.dtype, device=layer.codes.device).expand(layer.codes.shape)
            if layer.pq_packed_codes is not None:
                # scratch for the K-split launches (long-K layers, up to 128 tokens): fixed-size buffers per
                # device, created here, at load time, when a layer can use them at any batch size
                if any(ops.workspace_bytes(T, N, G, M, K, d, layer.codes.element_size(), layer.codebooks.dtype)
                       for T in (16, 128)):
                    ops.ensure_workspace(layer.codes.device)

    @staticmethod
    def _packed_operands(layer: torch.nn.Module):
        """(codes, codebooks) that go with layer.pq_packed_codes: the layer's own, or the folded pair."""
        folded = getattr(layer, "pq_folded", None)
        if folded is not None:
            return folded
        return layer.codes, layer.codebooks

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        codebooks = layer.codebooks
        parts = getattr(codebooks, "output_partition_sizes", [])
        packed = getattr(layer, "pq_packed_codes", None)
        if packed is not None:
            # the layer's launch arguments, validated once and again whenever a weight tensor has been replaced
            desc = getattr(layer, "pq_prepared", None)
            pc, pcb = self._packed_operands(layer)
            if desc is None or not desc.matches(packed, pcb.data, layer.scales.data):
                desc = layer.pq_prepared = ops.PackedLinear(packed, pc.data, pcb.data, layer.scales.data, parts)
            return ops.pq_gemm_packed_prepared(desc, x, bias)
        if getattr(layer, "pq_dequant_route", False) and x.numel() // x.shape[-1] > GENERIC_KERNEL_MAX_TOKENS:
            # the reference's order for this route (aqlm.py:143-156): without bias the GEMM output is scaled,
            # with bias the weight is scaled first
            w = ops.pq_dequant(layer.codes, codebooks, parts)
            if bias is None:
                out = F.linear(x, w)
                out *= layer.scales.view(-1)
                return out
            w *= layer.scales.view(-1, 1)
            return F.linear(x, w, bias)
        return ops.pq_gemm(x, layer.codes, codebooks, layer.scales, parts, bias)

    def apply_silu_and_mul(self, layer: torch.nn.Module, x: torch.Tensor,
                           bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`SiluAndMul()(self.apply(layer, x, bias))` for a gate/up layer (two equal output partitions), the pair
        LlamaMLP.forward runs back to back (vllm/model_executor/models/llama.py:89-93): the same values, with the
        activation in the GEMM's epilogue when the library fuses it (decode batches)."""
        parts = list(getattr(layer.codebooks, "output_partition_sizes", []))
        packed = getattr(layer, "pq_packed_codes", None)
        if packed is None or len(parts) != 2 or parts[0] != parts[1] or layer.codes.shape[2] != 1:
            y = self.apply(layer, x, bias)
            out = torch.empty(y.shape[:-1] + (y.shape[-1] // 2,), dtype=y.dtype, device=y.device)
            ops.silu_and_mul(out, y)
            return out
        pc, pcb = self._packed_operands(layer)
        return ops.pq_gemm_packed_silu_mul(x, packed, pc, pcb, layer.scales, bias)


    # ---- a fused add + RMSNorm between two PQ linears (decode batches, no tensor parallelism on the first) ----
    def norm_pair_supported(self, first: torch.nn.Module, second: torch.nn.Module, tokens: int) -> bool:
        """Whether `first` (the linear whose output a fused add + RMSNorm takes: o_proj, down_proj) and `second` (the
        linear behind the norm: gate_up_proj, qkv_proj) run a batch of `tokens` as apply_add_norm / apply_normed."""
        for layer in (first, second):
            if getattr(layer, "pq_packed_codes", None) is None:
                return False
        if getattr(first, "tp_size", 1) > 1 and getattr(first, "reduce_results", False):
            return False            # the sum over ranks comes between the GEMM and the add
        pf = getattr(first.codebooks, "output_partition_sizes", [])
        ps = getattr(second.codebooks, "output_partition_sizes", [])
        fc, fcb = self._packed_operands(first)
        sc, scb = self._packed_operands(second)
        return (first.codes.shape[0] == second.codes.shape[1] * second.codebooks.shape[3]
                and ops.gemm_add_norm_supported(tokens, fc, fcb, pf)
                and ops.gemm_normed_supported(tokens, sc, scb, ps))

    def norm_pair_preferred(self, first: torch.nn.Module, second: torch.nn.Module, tokens: int) -> bool:
        """norm_pair_supported, and the pair is not slower than the three separate steps: both GEMMs of the pair run
        on the decode kernel, which takes more than 16 tokens as groups of 16 that each gather the weights again.  Up
        to 48 tokens the two norm launches saved outweigh that on every layer size measured (32 Llama-3-8B layers:
        1.54 against 1.65 ms at 32 tokens, level at 48, 2.53 against 2.26 ms at 64); beyond, only where the plain
        dispatch picks the decode kernel for both layers anyway (layers whose blocks hold few units)."""
        if not self.norm_pair_supported(first, second, tokens):
            return False
        if tokens <= 48:
            return True
        for layer in (first, second):
            pc, cb = self._packed_operands(layer)
            N, G, M = pc.shape
            parts = list(getattr(layer.codebooks, "output_partition_sizes", [])) or [N]
            name = ops.gemm_kernel_name(tokens, N, G, M, cb.shape[1], cb.shape[3], pc.element_size(),
                                        cb.dtype, parts, packed=True)
            if name != "mfma_decode_packed":
                return False
        return True

    def apply_add_norm(self, layer: torch.nn.Module, x: torch.Tensor, residual: torch.Tensor,
                       norm_weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
        """apply() of the linear in front of `RMSNorm.forward(x, residual)` (vllm/model_executor/layers/layernorm.py:
        50-80 with the fused add): `residual` becomes apply(x) + residual in place; returns (xw, ssq) for apply_normed
        of the linear behind the norm."""
        parts = getattr(layer.codebooks, "output_partition_sizes", [])
        pc, pcb = self._packed_operands(layer)
        return ops.pq_gemm_packed_add_norm(x, layer.pq_packed_codes, pc, pcb, layer.scales, parts,
                                           bias, residual, norm_weight)

    def apply_normed(self, layer: torch.nn.Module, xw: torch.Tensor, ssq: torch.Tensor, eps: float,
                     bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """apply(norm(...)) of the linear behind the norm, from what apply_add_norm returned."""
        parts = getattr(layer.codebooks, "output_partition_sizes", [])
        pc, pcb = self._packed_operands(layer)
        return ops.pq_gemm_packed_normed(xw, ssq, eps, layer.pq_packed_codes, pc, pcb, layer.scales,
                                         parts, bias)

    def apply_normed_silu_and_mul(self, layer: torch.nn.Module, xw: torch.Tensor, ssq: torch.Tensor, eps: float,
                                  bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`SiluAndMul()(self.apply_normed(layer, ...))` for a gate/up layer: the same values, in one launch where
        the library fuses the activation into the GEMM behind the norm."""
        parts = list(getattr(layer.codebooks, "output_partition_sizes", []))
        tokens = xw.numel() // xw.shape[-1] if xw.shape[-1] else 0
        pc, pcb = self._packed_operands(layer)
        if (len(parts) == 2 and parts[0] == parts[1] and layer.codes.shape[2] == 1
                and ops.gemm_normed_silu_mul_supported(tokens, pc, pcb)):
            return ops.pq_gemm_packed_normed_silu_mul(xw, ssq, eps, layer.pq_packed_codes, pc, pcb,
                                                      layer.scales, bias)
        y = self.apply_normed(layer, xw, ssq, eps, bias)
        out = torch.empty(y.shape[:-1] + (y.shape[-1] // 2,), dtype=y.dtype, device=y.device)
        ops.silu_and_mul(out, y)
        return out


class PQEmbeddingMethod(PQLinearMethod):
    """PQ-compressed vocabulary table: [rows of this rank, embedding_dim] as one output partition.  apply() gives
    the LM head's logits (the linear method's fused GEMM), embedding() the rows of the reconstructed table picked
    by token ids -- bit for bit F.embedding over the dequantised table, which is never built."""

    def embedding(self, layer: torch.nn.Module, input_: torch.Tensor) -> torch.Tensor:
        parts = getattr(layer.codebooks, "output_partition_sizes", [])
        return ops.pq_embedding(input_, layer.codes, layer.codebooks, layer.scales, parts)


# batch size up to which a format without an MFMA kernel stays on the generic fused kernel (the reference switches
# from its gemv kernel to dequantise + GEMM above 6 tokens, aqlm.py:344-349)
GENERIC_KERNEL_MAX_TOKENS = 8


def register() -> None:
    """Idempotent registration of the "pq" method (entry point for vLLM's `vllm.general_plugins`
    group, which runs in every engine and worker process).  Also adds the name to the ROCm
    platform's allow-list, which otherwise rejects unknown methods (vllm/platforms/rocm.py:108-111)."""
    if PQ_METHOD_NAME not in QUANTIZATION_METHODS:
        register_quantization_config(PQ_METHOD_NAME)(PQConfig)
    try:
        from vllm.platforms.rocm import RocmPlatform
        if PQ_METHOD_NAME not in RocmPlatform.supported_quantization:
            RocmPlatform.supported_quantization.append(PQ_METHOD_NAME)
    except Exception:
        pass
