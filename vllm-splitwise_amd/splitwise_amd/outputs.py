"""Request outputs returned to callers (vllm/outputs.py:9-132)."""
import time
from typing import List, Optional

from .sequence import RequestMetrics, SequenceGroup, SequenceStatus


class CompletionOutput:
    def __init__(self, index: int, text: str, token_ids: List[int], cumulative_logprob: float,
                 logprobs=None, finish_reason: Optional[str] = None) -> None:
        self.index = index
        self.text = text
        self.token_ids = token_ids
        self.cumulative_logprob = cumulative_logprob
        self.logprobs = logprobs
        self.finish_reason = finish_reason

    def finished(self) -> bool:
        return self.finish_reason is not None

    def __repr__(self) -> str:
        return (f"CompletionOutput(index={self.index}, text={self.text!r}, token_ids={self.token_ids}, "
                f"cumulative_logprob={self.cumulative_logprob}, finish_reason={self.finish_reason})")


class RequestOutput:
    def __init__(self, request_id: str, prompt: Optional[str], prompt_token_ids: List[int], prompt_logprobs,
                 outputs: List[CompletionOutput], finished: bool,
                 metrics: Optional[RequestMetrics] = None) -> None:
        self.request_id = request_id
        self.prompt = prompt
        self.prompt_token_ids = prompt_token_ids
        self.prompt_logprobs = prompt_logprobs
        self.outputs = outputs
        self.finished = finished
        self.metrics = metrics

    @classmethod
    def from_seq_group(cls, seq_group: SequenceGroup) -> "RequestOutput":
        seqs = seq_group.get_seqs()
        n = seq_group.sampling_params.n
        want_logprobs = seq_group.sampling_params.logprobs is not None
        if len(seqs) == 1:                               # the common case, once per request and engine step
            s = seqs[0]
            outputs = [CompletionOutput(0, s.output_text, list(s.data.output_token_ids), s.data.cumulative_logprob,
                                        s.output_logprobs if want_logprobs else None,
                                        SequenceStatus.get_finished_reason(s.status))]
        else:
            if len(seqs) > n:
                seqs = sorted(seqs, key=lambda s: s.get_cumulative_logprob(), reverse=True)[:n]
            outputs = [CompletionOutput(seqs.index(s), s.output_text, list(s.get_output_token_ids()),
                                        s.get_cumulative_logprob(), s.output_logprobs if want_logprobs else None,
                                        SequenceStatus.get_finished_reason(s.status)) for s in seqs]
        finished = seq_group.is_finished()
        if finished:
            seq_group.set_finished_time(time.time())
        return cls(seq_group.request_id, seq_group.prompt, seq_group.prompt_token_ids, seq_group.prompt_logprobs,
                   outputs, finished, seq_group.metrics)

    def __repr__(self) -> str:
        return (f"RequestOutput(request_id={self.request_id}, prompt_token_ids={self.prompt_token_ids}, "
                f"outputs={self.outputs}, finished={self.finished})")
