"""Vocabulary-parallel embedding table and LM head (stand-alone mirror of
vllm/model_executor/layers/vocab_parallel_embedding.py), the two callers of a quantization method that are
not linear layers:

    UnquantizedEmbeddingMethod      :22-47
    pad_vocab_size                  :50-53
    VocabParallelEmbeddingShardIndices :75-136, _get_indices :271-298
    get_masked_input_and_mask       :139-156
    VocabParallelEmbedding          :159-430   (weight_loader :348-401, forward :403-422)
    ParallelLMHead                  :433-484
    compute_logits                  <-> LogitsProcessor._get_logits, layers/logits_processor.py:101-118

Same constructor arguments, shard layout (base vocabulary and added vocabulary padded and sharded separately,
added rows behind the base rows of every rank), loader behaviour (rows outside the checkpoint are zero) and
forward contract.  The table is [vocab rows per rank, embedding_dim]: to a quantization method it is a linear
weight with one output partition, so a PQ table uses the linear method's parameters unchanged.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from .base_config import QuantizationConfig, QuantizeMethodBase
from .parallel_state import (divide, get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                             tensor_model_parallel_all_gather, tensor_model_parallel_all_reduce)
from .utils import set_weight_attrs

DEFAULT_VOCAB_PADDING_SIZE = 64


class UnquantizedEmbeddingMethod(QuantizeMethodBase):

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        weight = Parameter(torch.empty(sum(output_partition_sizes), input_size_per_partition, dtype=params_dtype),
                           requires_grad=False)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0})
        layer.register_parameter("weight", weight)
        set_weight_attrs(weight, extra_weight_attrs)

    def apply(self, layer, x, bias=None):
        return F.linear(x, layer.weight, bias)

    def embedding(self, layer, input_):
        return F.embedding(input_, layer.weight)


def method_has_implemented_embedding(method_class) -> bool:
    """True when the class overrides QuantizeMethodBase.embedding (base_config.py:45-57)."""
    return getattr(method_class, "embedding", None) is not QuantizeMethodBase.embedding


def pad_vocab_size(vocab_size: int, pad_to: int = DEFAULT_VOCAB_PADDING_SIZE) -> int:
    return -(-vocab_size // pad_to) * pad_to


def _span(first: str, last: str) -> property:
    return property(lambda self: getattr(self, last) - getattr(self, first))


@dataclass
class VocabParallelEmbeddingShardIndices:
    """Token-id ranges of one rank.  `padded_*`: the rank's slice of the padded base / added vocabulary; the plain
    ones: the same slices cut at the real vocabulary sizes (what the rank actually holds)."""
    padded_org_vocab_start_index: int
    padded_org_vocab_end_index: int
    padded_added_vocab_start_index: int
    padded_added_vocab_end_index: int
    org_vocab_start_index: int
    org_vocab_end_index: int
    added_vocab_start_index: int
    added_vocab_end_index: int

    # row counts, with and without the padding
    num_org_elements = _span("org_vocab_start_index", "org_vocab_end_index")
    num_added_elements = _span("added_vocab_start_index", "added_vocab_end_index")
    num_org_elements_padded = _span("padded_org_vocab_start_index", "padded_org_vocab_end_index")
    num_added_elements_padded = _span("padded_added_vocab_start_index", "padded_added_vocab_end_index")
    num_org_vocab_padding = property(lambda self: self.num_org_elements_padded - self.num_org_elements)
    num_added_vocab_padding = property(lambda self: self.num_added_elements_padded - self.num_added_elements)
    num_elements_padded = property(lambda self: self.num_org_elements_padded + self.num_added_elements_padded)

    def __post_init__(self):
        for kind in ("org", "added"):
            lo, hi = getattr(self, f"{kind}_vocab_start_index"), getattr(self, f"{kind}_vocab_end_index")
            plo, phi = getattr(self, f"padded_{kind}_vocab_start_index"), getattr(self, f"padded_{kind}_vocab_end_index")
            if not (lo <= hi and plo <= phi and lo <= plo and hi <= phi and hi - lo <= phi - plo):
                raise AssertionError(f"inconsistent {kind} vocabulary ranges: {self}")


def get_masked_input_and_mask(input_: torch.Tensor, org_vocab_start_index: int, org_vocab_end_index: int,
                              num_org_vocab_padding: int, added_vocab_start_index: int,
                              added_vocab_end_index: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Token ids -> row numbers of this rank's table (0 for ids of other ranks) and the mask of those other ids."""
    in_org = (input_ >= org_vocab_start_index) & (input_ < org_vocab_end_index)
    in_added = (input_ >= added_vocab_start_index) & (input_ < added_vocab_end_index)
    # added rows sit behind the padded base rows
    added_shift = added_vocab_start_index - (org_vocab_end_index - org_vocab_start_index) - num_org_vocab_padding
    shift = org_vocab_start_index * in_org + added_shift * in_added
    mine = in_org | in_added
    return mine * (input_ - shift), ~mine


class VocabParallelEmbedding(torch.nn.Module):
    """Embedding table sharded over the vocabulary; forward sums the ranks' partial lookups."""

    def __init__(self, num_embeddings: int, embedding_dim: int, params_dtype: Optional[torch.dtype] = None,
                 org_num_embeddings: Optional[int] = None, padding_size: int = DEFAULT_VOCAB_PADDING_SIZE,
                 quant_config: Optional[QuantizationConfig] = None, prefix: str = ""):
        super().__init__()
        tp_rank = get_tensor_model_parallel_rank()
        self.tp_size = get_tensor_model_parallel_world_size()
        self.num_embeddings = num_embeddings
        self.padding_size = padding_size
        self.org_vocab_size = org_num_embeddings or num_embeddings
        num_added = num_embeddings - self.org_vocab_size
        self.org_vocab_size_padded = pad_vocab_size(self.org_vocab_size, padding_size)
        self.num_embeddings_padded = pad_vocab_size(self.org_vocab_size_padded + num_added, padding_size)
        assert self.org_vocab_size_padded <= self.num_embeddings_padded
        self.shard_indices = self._get_indices(self.num_embeddings_padded, self.org_vocab_size_padded,
                                               self.num_embeddings, self.org_vocab_size, tp_rank, self.tp_size)
        self.embedding_dim = embedding_dim

        quant_method = quant_config.get_quant_method(self, prefix=prefix) if quant_config is not None else None
        if quant_method is None:
            quant_method = UnquantizedEmbeddingMethod()
        # an embedding table needs embedding(); an LM head only apply()
        if type(self) is VocabParallelEmbedding and not method_has_implemented_embedding(type(quant_method)):
            raise NotImplementedError(f"The class {type(quant_method).__name__} must implement "
                                      "the 'embedding' method, see UnquantizedEmbeddingMethod.")
        self.quant_method: QuantizeMethodBase = quant_method

        if params_dtype is None:
            params_dtype = torch.get_default_dtype()
        self.num_added_embeddings = num_added
        self.num_embeddings_per_partition = divide(self.num_embeddings_padded, self.tp_size)
        assert self.shard_indices.num_elements_padded == self.num_embeddings_per_partition
        self.num_org_embeddings_per_partition = self.shard_indices.num_org_elements
        self.num_added_embeddings_per_partition = self.shard_indices.num_added_elements
        self.quant_method.create_weights(self, self.embedding_dim, [self.num_embeddings_per_partition],
                                         self.embedding_dim, self.num_embeddings_padded, params_dtype=params_dtype,
                                         weight_loader=self.weight_loader)

    @classmethod
    def _get_indices(cls, vocab_size_padded: int, org_vocab_size_padded: int, vocab_size: int, org_vocab_size: int,
                     tp_rank: int, tp_size: int) -> VocabParallelEmbeddingShardIndices:
        per_rank_org = divide(org_vocab_size_padded, tp_size)
        per_rank_added = divide(vocab_size_padded - org_vocab_size_padded, tp_size)
        p_org0, p_org1 = tp_rank * per_rank_org, (tp_rank + 1) * per_rank_org
        p_add0 = org_vocab_size + tp_rank * per_rank_added
        p_add1 = p_add0 + per_rank_added
        return VocabParallelEmbeddingShardIndices(
            p_org0, p_org1, p_add0, p_add1, min(p_org0, org_vocab_size), min(p_org1, org_vocab_size),
            min(p_add0, vocab_size), min(p_add1, vocab_size))

    def get_sharded_to_full_mapping(self) -> Optional[List[int]]:
        """Positions in the all-gathered logits, listed as: base ids in order, added ids in order, padding."""
        if self.tp_size < 2:
            return None
        base: List[int] = []
        added: List[int] = []
        padding: List[int] = []
        for r in range(self.tp_size):
            si = self._get_indices(self.num_embeddings_padded, self.org_vocab_size_padded, self.num_embeddings,
                                   self.org_vocab_size, r, self.tp_size)
            p = self.num_embeddings_per_partition * r
            base.extend(range(p, p + si.num_org_elements))
            padding.extend(range(p + si.num_org_elements, p + si.num_org_elements_padded))
            a0 = p + si.num_org_elements_padded
            added.extend(range(a0, a0 + si.num_added_elements))
            padding.extend(range(a0 + si.num_added_elements, a0 + si.num_added_elements_padded))
            assert a0 + si.num_added_elements_padded == p + self.num_embeddings_per_partition
        out = base + added + padding
        assert len(out) == self.num_embeddings_padded
        return out

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor):
        """This rank's rows of a checkpoint tensor of the whole (unpadded, base) vocabulary; the padding rows are
        zero.  A parameter without output_dim (a codebook) is replicated."""
        output_dim = getattr(param, "output_dim", None)
        if output_dim is None:
            assert param.data.shape == loaded_weight.shape
            param.data.copy_(loaded_weight)
            return
        start = self.shard_indices.org_vocab_start_index
        size = self.shard_indices.org_vocab_end_index - start
        packed_dim = getattr(param, "packed_dim", None)
        if packed_dim is not None and packed_dim == output_dim:
            pf = param.pack_factor
            assert loaded_weight.shape[output_dim] == self.org_vocab_size // pf
            start, size = start // pf, size // pf
        else:
            assert loaded_weight.shape[output_dim] == self.org_vocab_size
        shard = loaded_weight.narrow(output_dim, start, size)
        assert output_dim == 0
        param.data[:shard.shape[0]].copy_(shard)
        param.data[shard.shape[0]:].fill_(0)

    def forward(self, input_):
        if self.tp_size > 1:
            si = self.shard_indices
            masked_input, other = get_masked_input_and_mask(input_, si.org_vocab_start_index, si.org_vocab_end_index,
                                                            si.num_org_vocab_padding, si.added_vocab_start_index,
                                                            si.added_vocab_end_index)
        else:
            masked_input = input_
        out = self.quant_method.embedding(self, masked_input.long())
        if self.tp_size > 1:
            out.masked_fill_(other.unsqueeze(-1), 0)
        return tensor_model_parallel_all_reduce(out)

    def extra_repr(self) -> str:
        return (f"num_embeddings={self.num_embeddings_per_partition}, embedding_dim={self.embedding_dim}, "
                f"org_vocab_size={self.org_vocab_size}, num_embeddings_padded={self.num_embeddings_padded}, "
                f"tp_size={self.tp_size}")


class ParallelLMHead(VocabParallelEmbedding):
    """Output projection onto the vocabulary; its weights are used through quant_method.apply by the sampler's
    logits step (compute_logits below), never through forward."""

    def __init__(self, num_embeddings: int, embedding_dim: int, bias: bool = False,
                 params_dtype: Optional[torch.dtype] = None, org_num_embeddings: Optional[int] = None,
                 padding_size: int = DEFAULT_VOCAB_PADDING_SIZE, quant_config: Optional[QuantizationConfig] = None,
                 prefix: str = ""):
        super().__init__(num_embeddings, embedding_dim, params_dtype, org_num_embeddings, padding_size, quant_config,
                         prefix)
        self.quant_config = quant_config
        if bias:
            self.bias = Parameter(torch.empty(self.num_embeddings_per_partition, dtype=params_dtype))
            set_weight_attrs(self.bias, {"output_dim": 0, "weight_loader": self.weight_loader})
        else:
            self.register_parameter("bias", None)

    def tie_weights(self, embed_tokens: VocabParallelEmbedding):
        self.weight = embed_tokens.weight
        return self

    def forward(self, input_):
        del input_
        raise RuntimeError("LMHead's weights should be used in the sampler.")


def compute_logits(lm_head: VocabParallelEmbedding, hidden_states: torch.Tensor,
                   embedding_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """hidden_states [T, embedding_dim] -> logits [T, org_vocab_size]: the head's quantization method, an
    all-gather of the ranks' vocabulary shards, the padding cut off."""
    logits = lm_head.quant_method.apply(lm_head, hidden_states, bias=embedding_bias)
    logits = tensor_model_parallel_all_gather(logits)
    return logits[..., :lm_head.org_vocab_size]
