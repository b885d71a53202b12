"""Sequences, sequence groups and the per-step metadata handed to the workers.

Public names and method names follow vllm/sequence.py so code written against the
reference's data model keeps working; the bookkeeping itself is arithmetic (a sequence
knows how many blocks it spans, it does not carry per-block token objects).
"""
import copy
import enum
from dataclasses import dataclass, field
from typing import Dict, List, Optional

from .sampling_params import SamplingParams


@dataclass
class Logprob:
    logprob: float
    decoded_token: Optional[str] = None


class SequenceStatus(enum.Enum):
    WAITING = enum.auto()
    RUNNING = enum.auto()
    SWAPPED = enum.auto()
    FINISHED_STOPPED = enum.auto()
    FINISHED_LENGTH_CAPPED = enum.auto()
    FINISHED_ABORTED = enum.auto()
    FINISHED_IGNORED = enum.auto()

    @staticmethod
    def is_finished(status: "SequenceStatus") -> bool:
        return status in _FINISHED

    @staticmethod
    def get_finished_reason(status: "SequenceStatus") -> Optional[str]:
        return _FINISH_REASON.get(status)


_FINISHED = frozenset({SequenceStatus.FINISHED_STOPPED, SequenceStatus.FINISHED_LENGTH_CAPPED,
                       SequenceStatus.FINISHED_ABORTED, SequenceStatus.FINISHED_IGNORED})
# vllm/sequence.py:46-61: an ignored (too long) prompt reports "length", like the OpenAI API
_FINISH_REASON = {SequenceStatus.FINISHED_STOPPED: "stop", SequenceStatus.FINISHED_LENGTH_CAPPED: "length",
                  SequenceStatus.FINISHED_ABORTED: "abort", SequenceStatus.FINISHED_IGNORED: "length"}


@dataclass
class RequestMetrics:
    """vllm/sequence.py:64-80."""
    arrival_time: float
    last_token_time: float
    first_scheduled_time: Optional[float]
    first_token_time: Optional[float]
    time_in_queue: Optional[float]
    finished_time: Optional[float] = None


# Stands for a token that has been sampled on the device but has not reached the host yet
# (pipelined engine steps): it takes the token's place in the sequence until the value arrives.
PLACEHOLDER_TOKEN_ID = -1


class SequenceData:
    def __init__(self, prompt_token_ids: List[int], output_token_ids: Optional[List[int]] = None) -> None:
        self.prompt_token_ids = prompt_token_ids
        self.output_token_ids = [] if output_token_ids is None else output_token_ids
        self.cumulative_logprob = 0.0

    def append_token_id(self, token_id: int, logprob: float) -> None:
        self.output_token_ids.append(token_id)
        self.cumulative_logprob += logprob

    def get_len(self) -> int:
        return len(self.prompt_token_ids) + len(self.output_token_ids)

    def get_prompt_len(self) -> int:
        return len(self.prompt_token_ids)

    def get_output_len(self) -> int:
        return len(self.output_token_ids)

    def get_token_ids(self) -> List[int]:
        return self.prompt_token_ids + self.output_token_ids

    def get_last_token_id(self) -> int:
        return self.output_token_ids[-1] if self.output_token_ids else self.prompt_token_ids[-1]

    def __repr__(self) -> str:
        return f"SequenceData(prompt_len={len(self.prompt_token_ids)}, output_token_ids={self.output_token_ids})"


class Sequence:
    def __init__(self, seq_id: int, prompt: Optional[str], prompt_token_ids: List[int], block_size: int,
                 eos_token_id: Optional[int] = None) -> None:
        self.seq_id = seq_id
        self.prompt = prompt
        self.block_size = block_size
        self.eos_token_id = eos_token_id
        self.data = SequenceData(prompt_token_ids)
        self.output_logprobs: List[Dict[int, Logprob]] = []
        self.output_text = ""
        self.status = SequenceStatus.WAITING
        # incremental detokenisation state (engine-owned)
        self.prefix_offset = 0
        self.read_offset = 0
        self.tokens: Optional[List[str]] = None

    @property
    def num_logical_blocks(self) -> int:
        """Blocks spanned by the tokens seen so far (len(logical_token_blocks) in the reference)."""
        return -(-self.data.get_len() // self.block_size)

    def hash_of_block(self, logical_idx: int) -> int:
        """Content address of a block for prefix caching: every token from the start of the sequence to
        the block's end (vllm/sequence.py:188-194; the second member is the reference's lora_int_id)."""
        return hash((tuple(self.data.get_token_ids()[0:self.num_hashed_tokens_of_block(logical_idx)]), 0))

    def num_hashed_tokens_of_block(self, logical_idx: int) -> int:
        return logical_idx * self.block_size + self.block_size

    def append_token_id(self, token_id: int, logprobs: Dict[int, Logprob]) -> None:
        assert token_id in logprobs
        self.output_logprobs.append(logprobs)
        self.data.append_token_id(token_id, logprobs[token_id].logprob)

    def append_placeholder(self) -> None:
        self.output_logprobs.append({})
        self.data.output_token_ids.append(PLACEHOLDER_TOKEN_ID)

    def has_placeholder(self) -> bool:
        out = self.data.output_token_ids
        return bool(out) and out[-1] == PLACEHOLDER_TOKEN_ID

    def fill_placeholder(self, token_id: int, logprobs: Dict[int, Logprob]) -> None:
        assert self.has_placeholder() and token_id in logprobs
        self.data.output_token_ids[-1] = token_id
        self.output_logprobs[-1] = logprobs
        self.data.cumulative_logprob += logprobs[token_id].logprob

    def get_len(self) -> int:
        return self.data.get_len()

    def get_prompt_len(self) -> int:
        return self.data.get_prompt_len()

    def get_output_len(self) -> int:
        return self.data.get_output_len()

    def get_token_ids(self) -> List[int]:
        return self.data.get_token_ids()

    def get_last_token_id(self) -> int:
        return self.data.get_last_token_id()

    def get_output_token_ids(self) -> List[int]:
        return self.data.output_token_ids

    def get_cumulative_logprob(self) -> float:
        return self.data.cumulative_logprob

    def is_finished(self) -> bool:
        return self.status in _FINISHED

    def fork(self, new_seq_id: int) -> "Sequence":
        child = copy.deepcopy(self)
        child.seq_id = new_seq_id
        return child

    def __repr__(self) -> str:
        return f"Sequence(seq_id={self.seq_id}, status={self.status.name}, num_blocks={self.num_logical_blocks})"


@dataclass
class SequenceGroupState:
    generator: Optional[object] = None      # torch.Generator for seeded sampling


class SequenceGroup:
    def __init__(self, request_id: str, seqs: List[Sequence], sampling_params: SamplingParams,
                 arrival_time: float) -> None:
        self.request_id = request_id
        self.seqs_dict = {seq.seq_id: seq for seq in seqs}
        self.sampling_params = sampling_params
        self.metrics = RequestMetrics(arrival_time=arrival_time, last_token_time=arrival_time,
                                      first_scheduled_time=None, first_token_time=None, time_in_queue=None)
        self.prompt_logprobs = None
        self.state = SequenceGroupState()

    @property
    def prompt(self) -> Optional[str]:
        return next(iter(self.seqs_dict.values())).prompt

    @property
    def prompt_token_ids(self) -> List[int]:
        return next(iter(self.seqs_dict.values())).data.prompt_token_ids

    def get_last_latency(self, now: float) -> float:
        latency = now - self.metrics.last_token_time
        self.metrics.last_token_time = now
        return latency

    def maybe_set_first_token_time(self, time: float) -> None:
        if self.metrics.first_token_time is None:
            self.metrics.first_token_time = time

    def maybe_set_first_scheduled_time(self, time: float) -> None:
        if self.metrics.first_scheduled_time is None:
            self.metrics.first_scheduled_time = time
            self.metrics.time_in_queue = time - self.metrics.arrival_time

    def set_finished_time(self, time: Optional[float]) -> None:
        self.metrics.finished_time = time

    # (these run a few thousand times per engine step with hundreds of requests in flight, nearly all of them groups
    # of one sequence: that case takes no list comprehension)
    def get_max_num_running_seqs(self) -> int:
        """vllm/sequence.py:363-378 without the beam-search branch."""
        d = self.seqs_dict
        if len(d) == 1 and self.sampling_params.best_of == 1:
            for s in d.values():
                return 0 if s.status in _FINISHED else 1
        if self.sampling_params.best_of > self.num_seqs():
            return self.sampling_params.best_of
        return self.num_unfinished_seqs()

    def get_seqs(self, status: Optional[SequenceStatus] = None) -> List[Sequence]:
        d = self.seqs_dict
        if len(d) == 1:
            for s in d.values():
                return [s] if status is None or s.status == status else []
        if status is None:
            return list(d.values())
        return [s for s in d.values() if s.status == status]

    def get_unfinished_seqs(self) -> List[Sequence]:
        return [s for s in self.seqs_dict.values() if s.status not in _FINISHED]

    def get_finished_seqs(self) -> List[Sequence]:
        return [s for s in self.seqs_dict.values() if s.status in _FINISHED]

    def num_seqs(self, status: Optional[SequenceStatus] = None) -> int:
        d = self.seqs_dict
        if status is None:
            return len(d)
        if len(d) == 1:
            for s in d.values():
                return 1 if s.status == status else 0
        return sum(1 for s in d.values() if s.status == status)

    def num_unfinished_seqs(self) -> int:
        return sum(1 for s in self.seqs_dict.values() if s.status not in _FINISHED)

    def num_finished_seqs(self) -> int:
        return len(self.get_finished_seqs())

    def find(self, seq_id: int) -> Sequence:
        if seq_id not in self.seqs_dict:
            raise ValueError(f"Sequence {seq_id} not found.")
        return self.seqs_dict[seq_id]

    def add(self, seq: Sequence) -> None:
        if seq.seq_id in self.seqs_dict:
            raise ValueError(f"Sequence {seq.seq_id} already exists.")
        self.seqs_dict[seq.seq_id] = seq

    def remove(self, seq_id: int) -> None:
        if seq_id not in self.seqs_dict:
            raise ValueError(f"Sequence {seq_id} not found.")
        del self.seqs_dict[seq_id]

    def is_finished(self) -> bool:
        for s in self.seqs_dict.values():
            if s.status not in _FINISHED:
                return False
        return True

    def __repr__(self) -> str:
        return f"SequenceGroup(request_id={self.request_id}, num_seqs={len(self.seqs_dict)})"


class SequenceGroupMetadata:
    """What a worker needs to run one sequence group for one step (vllm/sequence.py:429-465)."""

    def __init__(self, request_id: str, is_prompt: bool, seq_data: Dict[int, SequenceData],
                 sampling_params: SamplingParams, block_tables: Optional[Dict[int, List[int]]],
                 computed_block_nums: Optional[List[int]] = None,
                 state: Optional[SequenceGroupState] = None) -> None:
        self.request_id = request_id
        self.is_prompt = is_prompt
        self.seq_data = seq_data
        self.sampling_params = sampling_params
        self.block_tables = block_tables
        self.computed_block_nums = computed_block_nums
        self.state = SequenceGroupState() if state is None else state


class SequenceOutput:
    def __init__(self, parent_seq_id: int, output_token: int, logprobs: Dict[int, Logprob]) -> None:
        self.parent_seq_id = parent_seq_id
        self.output_token = output_token
        self.logprobs = logprobs

    def __eq__(self, other: object) -> bool:
        return (isinstance(other, SequenceOutput) and self.parent_seq_id == other.parent_seq_id
                and self.output_token == other.output_token and self.logprobs == other.logprobs)

    def __repr__(self) -> str:
        return f"SequenceOutput(parent_seq_id={self.parent_seq_id}, output_token={self.output_token})"


class SequenceGroupOutput:
    def __init__(self, samples: List[SequenceOutput], prompt_logprobs=None) -> None:
        self.samples = samples
        self.prompt_logprobs = prompt_logprobs

    def __eq__(self, other: object) -> bool:
        return (isinstance(other, SequenceGroupOutput) and self.samples == other.samples
                and self.prompt_logprobs == other.prompt_logprobs)

    def __repr__(self) -> str:
        return f"SequenceGroupOutput(samples={self.samples})"


@dataclass
class SamplerOutput:
    """One SequenceGroupOutput per scheduled group, list-like (vllm/sequence.py:529-560)."""
    outputs: List[SequenceGroupOutput] = field(default_factory=list)
    # optional device tensors of a step (token probabilities, sampled ids) and a metrics slot: part of the
    # reference's data model, left None by this build's sampler
    sampled_token_probs: Optional[object] = None
    sampled_token_ids: Optional[object] = None
    spec_decode_worker_metrics: Optional[object] = None

    def __getitem__(self, idx: int) -> SequenceGroupOutput:
        return self.outputs[idx]

    def __setitem__(self, idx: int, value: SequenceGroupOutput) -> None:
        self.outputs[idx] = value

    def __len__(self) -> int:
        return len(self.outputs)

    def __eq__(self, other: object) -> bool:
        return isinstance(other, self.__class__) and self.outputs == other.outputs

    def reorder(self, positions: List[int]) -> None:
        """``positions[i]`` = where the i-th group of the caller's order sits in ``outputs`` (a step may run its
        groups in an order of its own)."""
        self.outputs = [self.outputs[p] for p in positions]

    def __iter__(self):
        return iter(self.outputs)


# ------------------------------------------------------------------ control-plane encodings
# A step command for a worker in another process is pickled.  A decode step reads only the length
# and the last token of a sequence, so the command carries those instead of every token id (a
# 256-sequence command shrinks from ~700 KB to ~60 KB, its pickling from ~17 ms to ~3 ms).
class DecodeSeqData:
    """What a decode step reads of a sequence (see ModelRunner._prepare_decode)."""
    __slots__ = ("length", "last_token_id")

    def __init__(self, length: int, last_token_id: int) -> None:
        self.length = length
        self.last_token_id = last_token_id

    def get_len(self) -> int:
        return self.length

    def get_last_token_id(self) -> int:
        return self.last_token_id

    def get_token_ids(self) -> List[int]:
        raise RuntimeError("a decode command does not carry the token ids of a sequence")


def slim_decode_metadata(metas: List[SequenceGroupMetadata]) -> List[SequenceGroupMetadata]:
    """The same decode step with each sequence reduced to (length, last token); the per-request
    sampling state stays with the worker (it is keyed by request id there)."""
    out = []
    for m in metas:
        if m.is_prompt:
            return metas                            # prompt steps need every token
        data = {sid: DecodeSeqData(d.get_len(), d.get_last_token_id()) for sid, d in m.seq_data.items()}
        out.append(SequenceGroupMetadata(m.request_id, False, data, m.sampling_params, m.block_tables))
    return out


def pack_sampler_output(output: "SamplerOutput") -> List[List[tuple]]:
    """[(sequence id, token, log-probability), ...] per group."""
    return [[(s.parent_seq_id, s.output_token, s.logprobs[s.output_token].logprob) for s in g.samples]
            for g in output.outputs]


def unpack_sampler_output(rows: List[List[tuple]]) -> "SamplerOutput":
    return SamplerOutput([SequenceGroupOutput([SequenceOutput(sid, tok, {tok: Logprob(lp)}) for sid, tok, lp in g],
                                              None) for g in rows])
