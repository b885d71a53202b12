"""Per-step attention metadata (field names of vllm/model_executor/input_metadata.py:6-54)."""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch


@dataclass
class InputMetadata:
    is_prompt: bool
    slot_mapping: torch.Tensor                    # int64; -1 = padding row, never cached
    prompt_lens: Optional[torch.Tensor]           # int32 [num_prompts] (prompt runs)
    max_seq_len: Optional[int]                    # longest prompt of the step
    start_loc: Optional[torch.Tensor]             # int32 [num_prompts]: first row of each prompt
    max_context_len: Optional[int]                # longest context of the step (decode runs)
    context_lens: Optional[torch.Tensor]          # int32 [num_seqs]
    block_tables: Optional[torch.Tensor]          # int32 [num_seqs, max_blocks]
    use_cuda_graph: bool
    kv_cache_dtype: str
    # {hand-off slot: [(first block, number of blocks), ...]} to push (prompt pool) or await (token pool);
    # a prompt pool with block numbers of its own pushes (first source block, first destination block, number)
    blocks_to_nw: Dict[int, List[Tuple[int, ...]]] = field(default_factory=dict)
    # the token pool that this prompt step's sequences decode on (prompt pools that feed several)
    nw_peer: int = 0
    # mean context of an eagerly run decode step (None: unknown, e.g. inside a captured graph)
    mean_context_len: Optional[float] = None
    # "v1" / "v2": the decode-attention kernel to use whatever the dispatch rule says (captured graphs exist in
    # both forms and the runner picks one per replay); None: the rule decides
    decode_kernel: Optional[str] = None
