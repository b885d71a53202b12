"""Sampling parameters (the subset of vllm/sampling_params.py the hot path reads)."""
import enum
from typing import List, Optional, Union

_EPS = 1e-5


class SamplingType(enum.IntEnum):
    GREEDY = 0
    RANDOM = 1
    BEAM = 2


class SamplingParams:
    """Field names and defaults follow vllm/sampling_params.py:19-143.  Beam search,
    penalties and logits processors are outside this build and rejected loudly."""

    def __init__(self, n: int = 1, best_of: Optional[int] = None, temperature: float = 1.0,
                 top_p: float = 1.0, top_k: int = -1, use_beam_search: bool = False,
                 stop: Optional[Union[str, List[str]]] = None, stop_token_ids: Optional[List[int]] = None,
                 ignore_eos: bool = False, max_tokens: Optional[int] = 16, logprobs: Optional[int] = None,
                 prompt_logprobs: Optional[int] = None, skip_special_tokens: bool = True,
                 seed: Optional[int] = None, include_stop_str_in_output: bool = False,
                 spaces_between_special_tokens: bool = True) -> None:
        self.n = n
        self.best_of = best_of if best_of is not None else n
        self.temperature = temperature
        self.top_p = top_p
        self.top_k = top_k
        self.use_beam_search = use_beam_search
        self.stop = [] if stop is None else ([stop] if isinstance(stop, str) else list(stop))
        self.stop_token_ids = [] if stop_token_ids is None else list(stop_token_ids)
        self.ignore_eos = ignore_eos
        self.max_tokens = max_tokens
        self.logprobs = logprobs
        self.prompt_logprobs = prompt_logprobs
        self.skip_special_tokens = skip_special_tokens
        self.seed = seed
        self.include_stop_str_in_output = include_stop_str_in_output
        self.spaces_between_special_tokens = spaces_between_special_tokens
        self._verify()

    def _verify(self) -> None:
        if self.n < 1:
            raise ValueError(f"n must be at least 1, got {self.n}.")
        if self.best_of < self.n:
            raise ValueError(f"best_of must be greater than or equal to n, got n={self.n} and "
                             f"best_of={self.best_of}.")
        if self.temperature < 0.0:
            raise ValueError(f"temperature must be non-negative, got {self.temperature}.")
        if not 0.0 < self.top_p <= 1.0:
            raise ValueError(f"top_p must be in (0, 1], got {self.top_p}.")
        if self.top_k < -1 or self.top_k == 0:
            raise ValueError(f"top_k must be -1 (disable), or at least 1, got {self.top_k}.")
        if self.max_tokens is not None and self.max_tokens < 1:
            raise ValueError(f"max_tokens must be at least 1, got {self.max_tokens}.")
        if self.logprobs is not None and self.logprobs < 0:
            raise ValueError(f"logprobs must be non-negative, got {self.logprobs}.")
        if self.use_beam_search:
            raise NotImplementedError("beam search is outside the scope of this build")
        if self.best_of != 1 and self.temperature < _EPS:
            raise ValueError(f"best_of must be 1 when using greedy sampling. Got {self.best_of}.")

    @property
    def sampling_type(self) -> SamplingType:
        if self.temperature < _EPS:
            return SamplingType.GREEDY
        return SamplingType.RANDOM

    def __repr__(self) -> str:
        return (f"SamplingParams(n={self.n}, best_of={self.best_of}, temperature={self.temperature}, "
                f"top_p={self.top_p}, top_k={self.top_k}, max_tokens={self.max_tokens}, "
                f"ignore_eos={self.ignore_eos}, stop_token_ids={self.stop_token_ids}, seed={self.seed})")
