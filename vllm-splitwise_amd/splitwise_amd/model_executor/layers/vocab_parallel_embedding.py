"""Token embedding and output head sharded over the vocabulary.

vllm/model_executor/layers/vocab_parallel_embedding.py:35-148: the vocabulary (padded to a multiple
of 64 per the reference's pad_vocab_size, :14-17) is cut into tensor_parallel_size contiguous
ranges; a rank holds the rows of its range.  Embedding: tokens outside the range look up row 0 and
their rows are zeroed, then the partial results are summed over the ranks (:98-111).  Output head:
each rank multiplies by its rows, giving the logits of its vocabulary range, which the sampler
gathers on the pool's first rank (layers/sampler.py:44-57).
"""
from typing import Optional

import torch
import torch.nn as nn

from ..parallel_state import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                              tensor_model_parallel_all_reduce)

DEFAULT_VOCAB_PADDING_SIZE = 64


def pad_vocab_size(vocab_size: int, pad_to: int = DEFAULT_VOCAB_PADDING_SIZE) -> int:
    return -(-vocab_size // pad_to) * pad_to


class VocabParallelEmbedding(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int, params_dtype: Optional[torch.dtype] = None) -> None:
        super().__init__()
        self.org_vocab_size = num_embeddings
        self.embedding_dim = embedding_dim
        self.tp_size = get_tensor_model_parallel_world_size()
        # one rank keeps the table as it is; sharding pads it so that every rank holds equally many rows
        self.num_embeddings_padded = num_embeddings if self.tp_size == 1 else pad_vocab_size(num_embeddings)
        assert self.num_embeddings_padded % self.tp_size == 0
        self.num_embeddings_per_partition = self.num_embeddings_padded // self.tp_size
        rank = get_tensor_model_parallel_rank()
        self.vocab_start_index = rank * self.num_embeddings_per_partition
        self.vocab_end_index = self.vocab_start_index + self.num_embeddings_per_partition
        self.weight = nn.Parameter(torch.empty(self.num_embeddings_per_partition, embedding_dim,
                                               dtype=params_dtype or torch.get_default_dtype()), requires_grad=False)

    # ---- placing the FULL [vocab, hidden] table into this rank's rows (vocab_parallel_embedding.py:83-96)
    def full_shape(self, param: torch.Tensor):
        return (self.org_vocab_size, self.embedding_dim)

    def weight_loader(self, param: torch.Tensor, full: torch.Tensor) -> None:
        assert tuple(full.shape) == (self.org_vocab_size, self.embedding_dim), tuple(full.shape)
        rows = max(0, min(self.vocab_end_index, self.org_vocab_size) - self.vocab_start_index)
        param.data.zero_()                                     # padding rows stay zero
        if rows:
            param.data[:rows].copy_(full[self.vocab_start_index:self.vocab_start_index + rows])

    def forward(self, input_ids: torch.Tensor) -> torch.Tensor:
        if self.tp_size == 1:
            return torch.nn.functional.embedding(input_ids, self.weight)
        outside = (input_ids < self.vocab_start_index) | (input_ids >= self.vocab_end_index)
        local = (input_ids - self.vocab_start_index).masked_fill(outside, 0)
        out = torch.nn.functional.embedding(local, self.weight)
        out.masked_fill_(outside.unsqueeze(-1), 0)
        return tensor_model_parallel_all_reduce(out)


class ParallelLMHead(VocabParallelEmbedding):
    """The output head: same sharding, used as ``hidden @ weight.T`` by the sampler
    (vocab_parallel_embedding.py:114-148)."""

    def forward(self, input_ids):
        raise RuntimeError("LMHead's weights should be used in the sampler.")
