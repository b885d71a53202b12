"""Randomised scenarios for the pipelined split engine (one prompt pool feeding one or two token pools) on CPU:
the real engine, schedulers and block managers over an in-process stand-in executor whose prompt steps finish a
random number of engine calls after their launch and whose decode samples arrive one call late, as on the device.
Requests arrive while others decode, some stop on a token, some are aborted at a random moment, and the caches are
small enough to force preemption.  Checked for every seed: no exception, every surviving request's tokens are the
reference chain, and when the engine has drained every block and every hand-off slot is back."""
import random

import pytest

import conftest  # noqa: F401
from split_fakes import VOCAB, next_token
from splitwise_amd import EngineArgs, LLMEngine, SamplingParams
from splitwise_amd.executor.executor_base import ExecutorBase
from splitwise_amd.sequence import PLACEHOLDER_TOKEN_ID, Logprob, SamplerOutput, SequenceGroupOutput, SequenceOutput
from splitwise_amd.utils import MAX_SLOT_IDS


def _sample(metas) -> SamplerOutput:
    outs = []
    for meta in metas:
        (seq_id, data), = meta.seq_data.items()
        ids = data.get_token_ids()
        if ids[-1] == PLACEHOLDER_TOKEN_ID:          # this step's own placeholder
            ids = ids[:-1]
        assert PLACEHOLDER_TOKEN_ID not in ids
        tok = next_token(ids)
        outs.append(SequenceGroupOutput([SequenceOutput(seq_id, tok, {tok: Logprob(-0.5)})]))
    return SamplerOutput(outs)


class _PromptPending:
    """Finished after ``polls`` looks at it (the engine looks once per call)."""

    def __init__(self, metas, polls: int) -> None:
        self.metas, self.polls = metas, polls

    def done(self) -> bool:
        self.polls -= 1
        return self.polls < 0

    def wait(self) -> None:
        self.polls = -1

    def resolve(self) -> SamplerOutput:
        return _sample(self.metas)


class RandomSplitExecutor(ExecutorBase):
    supports_pipelined_split = True

    def __init__(self, lanes: int, rng: random.Random, staging_blocks: int) -> None:
        self.rng = rng
        self.prompt_cache_blocks = staging_blocks
        self.prev = [None] * lanes
        self.before = [None] * lanes
        self.prompts_in_flight = 0
        self.max_prompts_in_flight = 0
        self.pendings = []

    def execute_model(self, *a, **k):
        raise AssertionError("unused")

    def execute_split(self, prompt_kwargs, decode_kwargs):
        raise AssertionError("the pipelined engine does not use the lock-stepped dispatch")

    def launch_prompt(self, kwargs, lane: int = 0):
        self.pendings = [p for p in self.pendings if p.polls >= 0]
        self.max_prompts_in_flight = max(self.max_prompts_in_flight, len(self.pendings) + 1)
        # a staged prompt cache: the step's blocks are the prompt pool's own, inside one half of its cache
        if self.prompt_cache_blocks:
            used = [b for m in kwargs["seq_group_metadata_list"] for t in m.block_tables.values() for b in t]
            half = self.prompt_cache_blocks // 2
            assert used and len(set(used)) == len(used) and max(used) // half == min(used) // half
            for slot, runs in kwargs["blocks_to_nw"].items():
                assert all(len(run) == 3 for run in runs) and slot in kwargs["nw_epochs"]
        pending = _PromptPending(kwargs["seq_group_metadata_list"], self.rng.randrange(0, 4))
        self.pendings.append(pending)
        return pending

    def decode_send(self, kwargs, lane: int = 0) -> None:
        self.before[lane], self.prev[lane] = self.prev[lane], (kwargs["seq_group_metadata_list"] if kwargs else None)

    def decode_collect(self, lane: int = 0):
        return _sample(self.before[lane]) if self.before[lane] is not None else None

    def check_health(self) -> None:
        pass


def _chain(prompt, n, stop):
    seq, out = list(prompt), []
    for _ in range(n):
        tok = next_token(seq)
        seq.append(tok)
        out.append(tok)
        if tok == stop:
            break
    return out


@pytest.mark.parametrize("lanes", [1, 2])
@pytest.mark.parametrize("seed", range(12))
def test_random_arrivals_stops_aborts_and_preemptions(lanes, seed):
    rng = random.Random(1000 * lanes + seed)
    blocks = rng.choice([10, 14, 24, 64])
    max_seqs = rng.choice([3, 4, 6])
    args = EngineArgs(model="llama-tiny", sep_prompt_token=True, max_num_seqs=max_seqs, max_model_len=128,
                      max_num_batched_tokens=128, swap_space=0.001, pipeline_steps=True, token_pools=lanes,
                      staged_prompt_cache=lanes > 1 or rng.random() < 0.5)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    cc.num_gpu_blocks, cc.num_cpu_blocks = blocks, 8
    staging = 2 * (128 // 16 + max_seqs) if pc.staged_prompt_cache else 0
    ex = RandomSplitExecutor(lanes, rng, staging)
    engine = LLMEngine(mc, cc, pc, sc, dc, executor=ex)
    assert engine.pipeline_split and len(engine.schedulers) == lanes

    requests = {}
    for i in range(rng.randrange(8, 20)):
        prompt = [rng.randrange(VOCAB) for _ in range(rng.choice([1, 2, 15, 16, 17, 31, 40, 60]))]
        n = rng.choice([1, 2, 5, 17, 30, 70 if len(prompt) <= 40 else 30])    # long decodes of short prompts outgrow the cache
        stop = _chain(prompt, n, None)[min(2, n - 1)] if rng.random() < 0.3 else None
        requests[f"r{i}"] = (prompt, n, stop, rng.randrange(0, 40))            # ..., arrival call
    aborts = {rid: arrival + rng.randrange(0, 12) for rid, (_, _, _, arrival) in requests.items() if rng.random() < 0.25}

    finished, aborted = {}, set()
    calls = 0
    while calls < 45 or engine.has_unfinished_requests():
        for rid, (prompt, n, stop, arrival) in requests.items():
            if arrival == calls:
                engine.add_request(rid, None, SamplingParams(temperature=0.0, max_tokens=n, ignore_eos=True,
                                                             stop_token_ids=[stop] if stop is not None else []),
                                   prompt_token_ids=prompt)
        for rid, when in aborts.items():
            if when == calls and rid not in finished:
                engine.abort_request(rid)
                aborted.add(rid)
        for out in engine.step():
            assert PLACEHOLDER_TOKEN_ID not in out.outputs[0].token_ids
            if out.finished:
                finished[out.request_id] = list(out.outputs[0].token_ids)
        calls += 1
        assert calls < 4000, "the engine does not drain"

    for rid, (prompt, n, stop, _) in requests.items():
        if rid in finished:
            assert finished[rid] == _chain(prompt, n, stop), rid
        else:
            assert rid in aborted, f"{rid} neither finished nor was aborted"
    assert not engine._prompt_queue and ex.max_prompts_in_flight <= 2
    if blocks == 10 and max_seqs == 6 and sum(n >= 70 for _, n, _, _ in requests.values()) >= 4 and not aborts:
        assert sum(getattr(sched, "num_preemptions", 0) for sched in engine.schedulers) > 0
    for sched in engine.schedulers:
        assert sched.block_manager.get_num_free_gpu_blocks() == blocks
        assert sched.seq_to_slot_mapper.num_free() == MAX_SLOT_IDS
        assert not sched.held_after_abort and not sched.prefilling
