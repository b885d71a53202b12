"""Next-token sampling.

Restates the parts of vllm/model_executor/layers/sampler.py the configurations use:
hidden states are pruned to the rows that are sampled (:135-140), logits = h @ E^T (:108-121),
temperature / top-k / top-p filtering (:79-99, :207-238), greedy = argmax of the log-probs
(:417), random = multinomial over the filtered distribution.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from ...sampling_params import SamplingParams, SamplingType
from ...sequence import Logprob, SamplerOutput, SequenceData, SequenceGroupOutput, SequenceOutput

_EPS = 1e-5


class SamplingMetadata:
    """What the sampler needs besides the hidden states (vllm/model_executor/sampling_metadata.py)."""

    def __init__(self, seq_groups: Optional[List[Tuple[List[int], SamplingParams]]],
                 seq_data: Optional[Dict[int, SequenceData]], prompt_lens: Optional[List[int]],
                 selected_token_indices: torch.Tensor, generators=None, perform_sampling: bool = True,
                 defer: bool = False, is_prompt: Optional[List[bool]] = None) -> None:
        self.seq_groups = seq_groups
        self.seq_data = seq_data
        self.prompt_lens = prompt_lens
        self.selected_token_indices = selected_token_indices
        self.generators = generators
        self.perform_sampling = perform_sampling
        self.defer = defer          # leave the sampled tokens on the device; the caller resolves them later
        # per sequence group: a prompt step?  (a prompt with best_of > 1 draws best_of tokens from its one row)
        self.is_prompt = is_prompt


def _host_to_device(values, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    """A small per-row parameter vector, copied without making the host wait for the stream."""
    t = torch.tensor(values, dtype=dtype)
    if device.type == "cuda":
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)


def _build_output(seq_groups, ids: List[int], chosen_lp: Optional[List[float]], top=None,
                  extra=None) -> SamplerOutput:
    """``top``: (log-probs, token ids) of the most likely tokens of every row, as nested lists, when some
    request asked for ``logprobs``; a row's dict holds the sampled token and its request's top-k
    (sampler.py:540-587 of the reference).  ``extra``: {row: (token ids, log-probs)} of the further samples a
    prompt with best_of > 1 draws from its row; they all name the prompt sequence as their parent
    (sampler.py:367-369)."""
    outputs: List[SequenceGroupOutput] = []
    row = 0
    for seq_ids, params in seq_groups:
        samples = []
        for seq_id in seq_ids:
            lp = 0.0 if chosen_lp is None else chosen_lp[row]
            drawn = [(ids[row], lp)]
            if extra is not None and row in extra:
                drawn += list(zip(*extra[row]))
            for tok_id, tok_lp in drawn:
                entry = {tok_id: Logprob(tok_lp)}
                if top is not None and params.logprobs:
                    for t_lp, tok in zip(top[0][row][:params.logprobs], top[1][row][:params.logprobs]):
                        entry.setdefault(tok, Logprob(t_lp))
                samples.append(SequenceOutput(seq_id, tok_id, entry))
            row += 1
        outputs.append(SequenceGroupOutput(samples, None))
    return SamplerOutput(outputs=outputs)


class PendingSamples:
    """The sampled tokens of a step that is still running: ``next_ids`` stays on the device (the
    next step reads its input tokens from it), a copy is on its way to pinned host memory, and
    ``resolve`` waits for that copy only."""

    def __init__(self, seq_groups, next_ids: torch.Tensor, chosen_lp: Optional[torch.Tensor], top=None) -> None:
        self.seq_groups = seq_groups
        self.next_ids = next_ids
        self._top = None if top is None else tuple(
            torch.empty(t.shape, dtype=t.dtype, pin_memory=t.is_cuda).copy_(t, non_blocking=True) for t in top)
        self.row_of: Dict[int, int] = {}
        for seq_ids, _ in seq_groups:
            for seq_id in seq_ids:
                self.row_of[seq_id] = len(self.row_of)
        self.event = None
        # events of other streams that the step's work is spread over (a prompt worker's KV pushes run on
        # their own stream): the step has finished when these have completed too
        self.extra_events = []
        self.after_sync = []         # callables run by resolve() once the step has finished on the device
        self._report_order = None    # see reorder()
        if next_ids.is_cuda:
            self._ids = torch.empty(next_ids.shape, dtype=next_ids.dtype, pin_memory=True)
            self._ids.copy_(next_ids, non_blocking=True)
            self._lp = None
            if chosen_lp is not None:
                self._lp = torch.empty(chosen_lp.shape, dtype=chosen_lp.dtype, pin_memory=True)
                self._lp.copy_(chosen_lp, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self._ids, self._lp = next_ids, chosen_lp

    def done(self) -> bool:
        """True once the step has finished on the device (resolve() will not block)."""
        return (self.event is None or self.event.query()) and all(e.query() for e in self.extra_events)

    def wait(self) -> None:
        if self.event is not None:
            self.event.synchronize()
        for event in self.extra_events:
            event.synchronize()

    def resolve(self) -> SamplerOutput:
        self.wait()
        checks, self.after_sync = self.after_sync, []
        for check in checks:         # e.g. the worker's look at its hand-off / all-reduce status words
            check()
        out = _build_output(self.seq_groups, self._ids.tolist(), None if self._lp is None else self._lp.tolist(),
                            None if self._top is None else tuple(t.tolist() for t in self._top))
        if self._report_order is not None:
            out.reorder(self._report_order)
        return out

    def reorder(self, positions: List[int]) -> None:
        """The step ran its sequence groups in another order than it was asked for: ``positions[i]`` is where
        group i of the request sits in ``seq_groups``.  ``resolve`` reports in the requested order; ``row_of``
        and ``next_ids`` keep the order the step ran in (the next step looks rows up by sequence id)."""
        self._report_order = positions


class FollowerSamples:
    """What a tensor-parallel rank that does not sample keeps of a pipelined step: the sampled token ids, received
    from its pool's first rank on the device, so that the next decode step can read its input tokens from them
    exactly as the first rank does from its ``PendingSamples``.  Nothing to report."""

    def __init__(self, seq_groups, next_ids: torch.Tensor) -> None:
        self.seq_groups = seq_groups
        self.next_ids = next_ids
        self.row_of: Dict[int, int] = {}
        for seq_ids, _ in seq_groups:
            for seq_id in seq_ids:
                self.row_of[seq_id] = len(self.row_of)
        self.after_sync = []
        self.extra_events = []

    def done(self) -> bool:
        return True

    def wait(self) -> None:
        pass

    def resolve(self) -> None:
        checks, self.after_sync = self.after_sync, []
        for check in checks:
            check()
        return None

    def reorder(self, positions: List[int]) -> None:
        pass                         # rows stay in the order the step ran in, which is what the next step looks up


class Sampler(nn.Module):
    def __init__(self, vocab_size: int, org_vocab_size: Optional[int] = None) -> None:
        super().__init__()
        self.vocab_size = vocab_size
        self.org_vocab_size = org_vocab_size or vocab_size

    def _get_logits(self, hidden_states: torch.Tensor, embedding: torch.Tensor,
                    embedding_bias: Optional[torch.Tensor]) -> torch.Tensor:
        logits = torch.matmul(hidden_states, embedding.t())
        if embedding_bias is not None:
            logits += embedding_bias
        # the output head is sharded over the vocabulary: every rank has the logits of its range, the
        # pool's first rank (the one that samples and reports) gets them all (sampler.py:44-57)
        from ..parallel_state import tensor_model_parallel_gather
        logits = tensor_model_parallel_gather(logits, dst=0)
        if logits is not None:
            logits = logits[:, :self.org_vocab_size]           # drop the padding of the sharded vocabulary
        return logits

    def forward(self, embedding: torch.Tensor, hidden_states: torch.Tensor,
                sampling_metadata: SamplingMetadata,
                embedding_bias: Optional[torch.Tensor] = None) -> Optional[SamplerOutput]:
        hidden_states = hidden_states.reshape(-1, hidden_states.shape[-1])
        hidden_states = hidden_states.index_select(0, sampling_metadata.selected_token_indices)
        logits = self._get_logits(hidden_states, embedding, embedding_bias)
        if logits is None or not sampling_metadata.perform_sampling:
            return None                                        # a tensor-parallel follower: rank 0 samples
        return self.sample_from_logits(logits, sampling_metadata)

    @staticmethod
    def _apply_top_k_top_p(logits: torch.Tensor, top_ps: torch.Tensor, top_ks: torch.Tensor) -> torch.Tensor:
        # ascending sort; mask everything below the k-th largest, then the low-probability tail
        logits_sort, logits_idx = logits.sort(dim=-1, descending=False)
        top_k_mask = logits_sort.size(1) - top_ks.to(torch.long)
        top_k_mask = logits_sort.gather(1, top_k_mask.unsqueeze(dim=1))
        logits_sort.masked_fill_(logits_sort < top_k_mask, -float("inf"))
        probs_sum = logits_sort.softmax(dim=-1).cumsum(dim=-1)
        top_p_mask = probs_sum <= 1 - top_ps.unsqueeze(dim=1)
        top_p_mask[:, -1] = False                      # at least one token stays
        logits_sort.masked_fill_(top_p_mask, -float("inf"))
        return torch.empty_like(logits_sort).scatter_(dim=-1, index=logits_idx, src=logits_sort)

    def sample_from_logits(self, logits: torch.Tensor, md: SamplingMetadata) -> SamplerOutput:
        logits = logits.float()
        rows: List[SamplingParams] = []
        for seq_ids, params in md.seq_groups:
            if params.prompt_logprobs is not None:
                raise NotImplementedError("prompt_logprobs is outside the scope of this build")
            rows.extend([params] * len(seq_ids))
        assert len(rows) == logits.shape[0], (len(rows), logits.shape)
        all_greedy = all(p.sampling_type == SamplingType.GREEDY for p in rows)
        want_logprobs = any(p.logprobs is not None for p in rows)
        top = None
        if all_greedy and not want_logprobs:
            # fast path: argmax of the logits equals argmax of the log-probs; the sampled token's
            # log-probability is its logit minus the row's log-sum-exp (sampler.py:417 of the reference
            # takes it from the full log-softmax)
            best, next_ids = torch.max(logits, dim=-1)
            chosen_lp = best - torch.logsumexp(logits, dim=-1)
        else:
            temps = [p.temperature if p.temperature >= _EPS else 1.0 for p in rows]
            if any(t != 1.0 for t in temps):
                logits = logits / _host_to_device(temps, torch.float32, logits.device).unsqueeze(1)
            vocab = logits.shape[1]
            if any(p.top_p < 1.0 - _EPS or (p.top_k != -1 and p.top_k != vocab) for p in rows):
                top_ps = _host_to_device([p.top_p for p in rows], torch.float32, logits.device)
                top_ks = _host_to_device([vocab if p.top_k == -1 else min(p.top_k, vocab) for p in rows],
                                         torch.int32, logits.device)
                logits = self._apply_top_k_top_p(logits, top_ps, top_ks)
            probs = torch.softmax(logits, dim=-1)
            logprobs = torch.log_softmax(logits, dim=-1)
            greedy_ids = torch.argmax(logprobs, dim=-1)
            if all_greedy:
                next_ids = greedy_ids
            else:
                random_ids = torch.multinomial(probs, 1).squeeze(1)
                # rows of seeded requests redraw from their own generator (sampler.py:370-383)
                for row_idx, g in (md.generators or {}).items():
                    random_ids[row_idx] = torch.multinomial(probs[row_idx], 1, generator=g)[0]
                is_greedy = _host_to_device([p.sampling_type == SamplingType.GREEDY for p in rows], torch.bool,
                                            logits.device)
                next_ids = torch.where(is_greedy, greedy_ids, random_ids)
            chosen_lp = logprobs.gather(1, next_ids.unsqueeze(1)).squeeze(1)
            most = max((p.logprobs or 0) for p in rows)
            if most > 0:
                top = torch.topk(logprobs, min(most, logprobs.shape[1]), dim=-1)
                top = (top.values, top.indices)
        # a prompt with best_of > 1: best_of - 1 further independent draws from its row (the reference draws
        # best_of columns of one multinomial call, sampler.py:356-369)
        extra = None
        if md.is_prompt is not None and not all_greedy:
            row = 0
            for (seq_ids, params), prompt in zip(md.seq_groups, md.is_prompt):
                if prompt and params.best_of > 1:
                    gen = (md.generators or {}).get(row)
                    more = torch.multinomial(probs[row], params.best_of - 1, replacement=True, generator=gen)
                    extra = extra or {}
                    extra[row] = (more, logprobs[row].gather(0, more))
                row += len(seq_ids)
        if md.defer:
            if extra is not None:
                raise NotImplementedError("best_of > 1 cannot leave its samples on the device (pipelined steps)")
            return PendingSamples(md.seq_groups, next_ids, chosen_lp, top)
        # the one device -> host sync of the step
        if extra is not None:
            extra = {row: (toks.tolist(), lps.tolist()) for row, (toks, lps) in extra.items()}
        return _build_output(md.seq_groups, next_ids.tolist(), None if chosen_lp is None else chosen_lp.tolist(),
                             None if top is None else (top[0].tolist(), top[1].tolist()), extra)
