"""Pipelined engine steps on CPU: the real engine and scheduler over a stand-in executor whose
deferred samples behave like the device's (a step's token is a function of everything before it,
and becomes known to the host one engine call later).  The pipelined loop must produce exactly
the outputs of the plain loop, including under preemption, stop tokens and aborts."""
import pytest

import conftest  # noqa: F401
from split_fakes import VOCAB, next_token
from splitwise_amd import EngineArgs, LLMEngine, SamplingParams
from splitwise_amd.executor.executor_base import ExecutorBase
from splitwise_amd.sequence import (PLACEHOLDER_TOKEN_ID, Logprob, SamplerOutput, SequenceGroupOutput,
                                    SequenceOutput)


def _sample(metas) -> SamplerOutput:
    outs = []
    for meta in metas:
        (seq_id, data), = meta.seq_data.items()
        ids = data.get_token_ids()
        if ids[-1] == PLACEHOLDER_TOKEN_ID:        # this step's own placeholder: the step ran before it existed
            ids = ids[:-1]
        assert PLACEHOLDER_TOKEN_ID not in ids
        tok = next_token(ids)
        outs.append(SequenceGroupOutput([SequenceOutput(seq_id, tok, {tok: Logprob(-0.5)})]))
    return SamplerOutput(outs)


class _Deferred:
    def __init__(self, metas):
        self.metas = metas

    def resolve(self) -> SamplerOutput:
        return _sample(self.metas)


class FakeExecutor(ExecutorBase):
    supports_deferred_samples = True

    def __init__(self):
        self.steps = 0
        self.prompt_tokens_seen = []

    def execute_model(self, seq_group_metadata_list, blocks_to_swap_in, blocks_to_swap_out, blocks_to_copy,
                      blocks_to_nw=None, defer=False):
        self.steps += 1
        for meta in seq_group_metadata_list:
            (seq_id, data), = meta.seq_data.items()
            ids = data.get_token_ids()
            if meta.is_prompt:
                assert PLACEHOLDER_TOKEN_ID not in ids          # a prompt step needs every token on the host
            else:
                assert PLACEHOLDER_TOKEN_ID not in ids[:-1]     # at most the last token is still on the device
        return _Deferred(seq_group_metadata_list) if defer else _sample(seq_group_metadata_list)

    def check_health(self) -> None:
        pass


def _engine(pipeline: bool, blocks: int = 64, max_num_seqs: int = 4) -> LLMEngine:
    args = EngineArgs(model="llama-tiny", max_num_seqs=max_num_seqs, max_model_len=256, max_num_batched_tokens=256,
                      swap_space=0.001, pipeline_steps=pipeline)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    cc.num_gpu_blocks, cc.num_cpu_blocks = blocks, 8
    return LLMEngine(mc, cc, pc, sc, dc, executor=FakeExecutor())


def _run(engine, requests, late=None, abort_at=None):
    """-> {request id: (token ids, finish reason, cumulative logprob, number of outputs seen)}"""
    for rid, (prompt, params) in requests.items():
        engine.add_request(rid, None, params, prompt_token_ids=prompt)
    final, seen = {}, {}
    calls = 0
    while engine.has_unfinished_requests():
        calls += 1
        assert calls < 5000
        if late and calls in late:
            rid, (prompt, params) = late[calls]
            engine.add_request(rid, None, params, prompt_token_ids=prompt)
        if abort_at and calls == abort_at[0]:
            engine.abort_request(abort_at[1])
        for out in engine.step():
            seen[out.request_id] = seen.get(out.request_id, 0) + 1
            c = out.outputs[0]
            assert PLACEHOLDER_TOKEN_ID not in c.token_ids
            if out.finished:
                final[out.request_id] = (list(c.token_ids), c.finish_reason, round(c.cumulative_logprob, 6),
                                         seen[out.request_id])
    return final


def _requests(n, max_tokens, **kw):
    return {f"r{i}": ([(5 * i + j) % VOCAB for j in range(3 + 7 * i)],
                      SamplingParams(temperature=0.0, max_tokens=max_tokens + i, ignore_eos=True, **kw))
            for i in range(n)}


def test_pipelined_outputs_equal_plain_outputs():
    plain = _run(_engine(False), _requests(7, 9))
    piped = _run(_engine(True), _requests(7, 9))
    assert piped == plain
    assert all(len(v[0]) == 9 + i and v[1] == "length" and v[3] == 9 + i for i, v in
               ((int(k[1:]), v) for k, v in piped.items()))


def test_pipelined_with_preemption_by_recomputation():
    # 10 blocks of 16 tokens for four sequences that grow to ~60 tokens each: some are recomputed
    reqs = {f"r{i}": ([(3 * i + j) % VOCAB for j in range(20 + i)],
                      SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True)) for i in range(4)}
    e_plain, e_piped = _engine(False, blocks=10), _engine(True, blocks=10)
    plain, piped = _run(e_plain, reqs), _run(e_piped, reqs)
    assert e_plain.scheduler.num_preemptions > 0 and e_piped.scheduler.num_preemptions > 0
    assert {k: v[:3] for k, v in piped.items()} == {k: v[:3] for k, v in plain.items()}
    assert e_piped.scheduler.block_manager.get_num_free_gpu_blocks() == 10


def test_pipelined_stop_token_drops_the_speculative_step():
    base = _run(_engine(False), _requests(3, 30))
    stops = {}
    for rid, (toks, _, _, _) in base.items():
        stops[rid] = toks[7]                     # a token each request is known to produce
    def with_stops():
        reqs = _requests(3, 30)
        return {rid: (p, SamplingParams(temperature=0.0, max_tokens=sp.max_tokens, ignore_eos=True,
                                        stop_token_ids=[stops[rid]])) for rid, (p, sp) in reqs.items()}
    e_piped = _engine(True)
    plain, piped = _run(_engine(False), with_stops()), _run(e_piped, with_stops())
    assert piped == plain
    assert all(v[1] == "stop" and v[0][-1] == stops[k] for k, v in piped.items())
    assert e_piped.scheduler.block_manager.get_num_free_gpu_blocks() == 64


def test_pipelined_late_arrivals_and_abort():
    late = {4: ("late0", ([9, 8, 7, 6], SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True))),
            9: ("late1", ([1] * 40, SamplingParams(temperature=0.0, max_tokens=3, ignore_eos=True)))}
    plain = _run(_engine(False), _requests(3, 12), late=late)
    piped = _run(_engine(True), _requests(3, 12), late=late)
    assert {k: v[:3] for k, v in piped.items()} == {k: v[:3] for k, v in plain.items()}
    engine = _engine(True)
    out = _run(engine, _requests(3, 12), abort_at=(5, "r1"))
    assert "r1" not in out and set(out) == {"r0", "r2"}
    assert out["r0"][:3] == plain["r0"][:3] and out["r2"][:3] == plain["r2"][:3]
    assert engine.scheduler.block_manager.get_num_free_gpu_blocks() == 64
    assert not engine.has_unfinished_requests()


class _PerSequenceExecutor(FakeExecutor):
    """Sampling with temperature: what a sequence gets next depends on the sequence, not only on its tokens."""

    @staticmethod
    def _sample(metas) -> SamplerOutput:
        outs = []
        for meta in metas:
            (seq_id, data), = meta.seq_data.items()
            ids = [t for t in data.get_token_ids() if t != PLACEHOLDER_TOKEN_ID]
            tok = (next_token(ids) + 7 * seq_id) % VOCAB
            outs.append(SequenceGroupOutput([SequenceOutput(seq_id, tok, {tok: Logprob(-0.5)})]))
        return SamplerOutput(outs)

    def execute_model(self, seq_group_metadata_list, blocks_to_swap_in, blocks_to_swap_out, blocks_to_copy,
                      blocks_to_nw=None, defer=False):
        if not defer:
            return self._sample(seq_group_metadata_list)
        pending = _Deferred(seq_group_metadata_list)
        pending.resolve = lambda: self._sample(seq_group_metadata_list)
        return pending


def _blocks_shared_with_different_content(engine, block_size=16):
    bm = engine.scheduler.block_manager
    seqs = [g.get_seqs()[0] for g in engine.scheduler.running]
    found = []
    for i, a in enumerate(seqs):
        for b in seqs[i + 1:]:
            tables = bm.block_tables.get(a.seq_id, []), bm.block_tables.get(b.seq_id, [])
            for k, (x, y) in enumerate(zip(*tables)):
                if x != y:
                    continue
                ta, tb = ([t for t in s.get_token_ids()[block_size * k:block_size * (k + 1)]
                           if t != PLACEHOLDER_TOKEN_ID] for s in (a, b))
                n = min(len(ta), len(tb))
                if ta[:n] != tb[:n]:
                    found.append((k, x, a.seq_id, b.seq_id))
    return found


@pytest.mark.parametrize("prompt_len", [31, 32, 33, 47])
def test_pipelined_prefix_caching_never_hashes_a_token_that_is_not_there_yet(prompt_len):
    """The same prompt three times, sampled differently: a block that fills up while its newest token is still
    on the device must not be addressed by a hash over the placeholder (all three would get one block)."""
    def engine_for(pipeline):
        args = EngineArgs(model="llama-tiny", max_num_seqs=4, max_model_len=256, max_num_batched_tokens=256,
                          swap_space=0.001, pipeline_steps=pipeline, enable_prefix_caching=True)
        mc, cc, pc, sc, dc = args.create_engine_configs()
        cc.num_gpu_blocks, cc.num_cpu_blocks = 64, 8
        return LLMEngine(mc, cc, pc, sc, dc, executor=_PerSequenceExecutor())

    prompt = [(3 * j + 1) % VOCAB for j in range(prompt_len)]
    results = {}
    for pipeline in (False, True):
        engine = engine_for(pipeline)
        for i in range(3):
            engine.add_request(f"r{i}", None, SamplingParams(temperature=1.0, max_tokens=40, ignore_eos=True),
                               prompt_token_ids=prompt)
        final, wrong = {}, []
        while engine.has_unfinished_requests():
            for out in engine.step():
                if out.finished:
                    final[out.request_id] = list(out.outputs[0].token_ids)
            if pipeline:
                wrong += _blocks_shared_with_different_content(engine)
        assert not wrong, wrong
        results[pipeline] = final
        bm = engine.scheduler.block_manager
        assert bm.get_num_free_gpu_blocks() == 64 and not bm._postponed
    assert results[True] == results[False]
    assert len({tuple(v) for v in results[True].values()}) == 3


def test_a_block_filled_under_a_placeholder_becomes_shareable_once_its_token_is_known():
    args = EngineArgs(model="llama-tiny", max_num_seqs=4, max_model_len=256, max_num_batched_tokens=256,
                      swap_space=0.001, pipeline_steps=True, enable_prefix_caching=True)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    cc.num_gpu_blocks, cc.num_cpu_blocks = 64, 8
    engine = LLMEngine(mc, cc, pc, sc, dc, executor=FakeExecutor())
    bm = engine.scheduler.block_manager
    prompt = [(3 * j + 1) % VOCAB for j in range(28)]
    engine.add_request("a", None, SamplingParams(temperature=0.0, max_tokens=30, ignore_eos=True),
                       prompt_token_ids=prompt)
    seq = engine.scheduler.waiting[0].get_seqs()[0]
    while seq.get_len() < 40:
        engine.step()
    second = bm.block_tables[seq.seq_id][1]
    ids = seq.get_token_ids()
    assert PLACEHOLDER_TOKEN_ID not in ids[:32]
    # a prompt that starts with those 32 tokens finds the block (filled by decode steps) under its content
    assert bm.gpu_allocator.block_hash[second] == seq.hash_of_block(1)
    engine.add_request("b", None, SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True),
                       prompt_token_ids=ids[:32] + [5, 6, 7])
    while not any(g.request_id == "b" for g in engine.scheduler.running):
        engine.step()
    other = next(g for g in engine.scheduler.running if g.request_id == "b").get_seqs()[0]
    assert bm.block_tables[other.seq_id][:2] == bm.block_tables[seq.seq_id][:2]
    while engine.has_unfinished_requests():
        engine.step()
    assert bm.get_num_free_gpu_blocks() == 64


# ------------------------------------------------------------------ pipelined split pools
class _PromptPending:
    """A prompt step that stays 'running on the prompt GPU' until the test lets it finish."""

    def __init__(self, metas):
        self.metas = metas
        self.finished = False

    def done(self) -> bool:
        return self.finished

    def wait(self) -> None:
        self.finished = True

    def resolve(self) -> SamplerOutput:
        return _sample(self.metas)


class FakeSplitExecutor(ExecutorBase):
    supports_pipelined_split = True

    def __init__(self):
        self.prompt_pending = None
        self.decode_prev = None

    def execute_split(self, prompt_kwargs, decode_kwargs):
        raise AssertionError("the pipelined engine does not use the lock-stepped dispatch")

    def execute_model(self, *a, **k):
        raise AssertionError("unused")

    def launch_prompt(self, kwargs):
        self.prompt_pending = _PromptPending(kwargs["seq_group_metadata_list"])
        return self.prompt_pending

    def decode_send(self, kwargs):
        self.decode_before, self.decode_prev = self.decode_prev, (kwargs["seq_group_metadata_list"] if kwargs else None)

    def decode_collect(self):
        return _sample(self.decode_before) if self.decode_before is not None else None

    def check_health(self) -> None:
        pass


def test_abort_during_a_running_prompt_step_keeps_its_blocks_until_the_step_has_resolved():
    """The prompt GPU writes and pushes KV into the token GPU's blocks on its own clock: blocks of a
    sequence aborted while its prompt step runs must not be handed to another sequence before then."""
    args = EngineArgs(model="llama-tiny", sep_prompt_token=True, max_num_seqs=4, max_model_len=256,
                      max_num_batched_tokens=256, swap_space=0.001, pipeline_steps=True)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    cc.num_gpu_blocks, cc.num_cpu_blocks = 8, 8
    ex = FakeSplitExecutor()
    engine = LLMEngine(mc, cc, pc, sc, dc, executor=ex)
    assert engine.pipeline_split
    bm = engine.scheduler.block_manager
    params = SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True)
    engine.add_request("a", None, params, prompt_token_ids=list(range(40)))       # 3 blocks
    engine.step()                                                                  # its prompt step is launched
    pending = ex.prompt_pending
    assert pending is not None and not pending.done()
    seq_a = engine.scheduler.running[0].get_seqs()[0]
    blocks_a = set(bm.get_block_table(seq_a))
    assert len(blocks_a) == 3 and bm.get_num_free_gpu_blocks() == 5
    engine.abort_request("a")                                                      # e.g. the client went away
    assert bm.get_num_free_gpu_blocks() == 5, "blocks of a running prompt step were freed by the abort"
    # new work arrives and is scheduled while the aborted prompt still runs: it must get other blocks
    engine.add_request("b", None, params, prompt_token_ids=list(range(100, 140)))
    engine.step()
    second = ex.prompt_pending
    assert second is not pending, "a second prompt step is queued behind the one that still runs"
    seq_b = [g for g in engine.scheduler.running if g.request_id == "b"][0].get_seqs()[0]
    blocks_b = set(bm.get_block_table(seq_b))
    assert len(blocks_b) == 3 and not (blocks_b & blocks_a) and bm.get_num_free_gpu_blocks() == 2
    # a third prompt has to wait: two prompt steps are in flight.  The call has nothing else to do, so it waits
    # for the older one (a's), and what has finished by the end of a call is collected in that call: a's result
    # is dropped and only now do its blocks come back
    engine.add_request("c", None, params, prompt_token_ids=list(range(200, 210)))
    outs = engine.step()
    assert all(o.request_id != "a" for o in outs)
    assert ex.prompt_pending is second and pending.finished and bm.get_num_free_gpu_blocks() == 5
    engine.step()                                                                  # c is launched
    assert ex.prompt_pending is not second
    assert bm.get_num_free_gpu_blocks() == 8 - 3 - 1                               # a's blocks came back; b and c hold theirs
    second.finished = True
    ex.prompt_pending.finished = True
    final = {}
    for _ in range(50):
        for o in engine.step():
            if o.finished:
                final[o.request_id] = list(o.outputs[0].token_ids)
        if ex.prompt_pending is not None:
            ex.prompt_pending.finished = True
        if not engine.has_unfinished_requests():
            break
    want = {}
    for rid, prompt in (("b", list(range(100, 140))), ("c", list(range(200, 210)))):
        seq = list(prompt)
        for _ in range(4):
            seq.append(next_token(seq))
        want[rid] = seq[len(prompt):]
    assert final == want
    assert bm.get_num_free_gpu_blocks() == 8


@pytest.mark.parametrize("blocks", [64, 2])
def test_overlap_executor_runs_the_split_loop_over_one_worker(blocks, tmp_path, monkeypatch):
    """``overlap_prompt_decode`` on CPU: the engine's pipelined split loop over executor/overlap_executor.py with a
    stand-in worker (no streams: the two sides' steps run in turn).  The stand-in computes every token from what its
    cache holds and looks placeholders up in the samples of the last DECODE step, so a prompt step that disturbed
    them, a decode step ahead of its prompt or a block handed out too early gives a wrong token.  The small cache
    (the stand-in offers 16 blocks more than asked for: 18) forces preemption and recomputation."""
    import random

    from split_fakes import VOCAB, FakeWorker, next_token
    from splitwise_amd import EngineArgs, LLMEngine, SamplingParams
    from splitwise_amd.executor.overlap_executor import OverlapExecutor
    monkeypatch.setenv("SPLIT_FAKE_DIR", str(tmp_path))
    monkeypatch.setenv("SPLIT_FAKE_BLOCKS", str(blocks))
    args = EngineArgs(model="llama-tiny", max_num_seqs=6, max_model_len=128, max_num_batched_tokens=512,
                      swap_space=0.001, pipeline_steps=True, overlap_prompt_decode=True)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    assert pc.sep_prompt_token and pc.one_gpu_overlap and pc.world_size == 2
    executor = OverlapExecutor(mc, cc, pc, sc, dc, worker_cls=FakeWorker)
    engine = LLMEngine(mc, cc, pc, sc, dc, executor=executor)
    assert engine.pipeline_split and executor.driver_worker.mixed
    rng = random.Random(1)
    prompts = {str(i): [rng.randrange(VOCAB) for _ in range(rng.choice([1, 5, 16, 17, 40, 60]))] for i in range(14)}
    lens = {rid: rng.choice([1, 2, 9, 30, 40]) for rid in prompts}
    pending, finished, steps = list(prompts), {}, 0
    while pending or engine.has_unfinished_requests():
        if pending and steps % 2 == 0:
            rid = pending.pop(0)
            engine.add_request(rid, None, SamplingParams(temperature=0.0, max_tokens=lens[rid], ignore_eos=True),
                               prompt_token_ids=prompts[rid])
        for out in engine.step():
            if out.finished:
                finished[out.request_id] = list(out.outputs[0].token_ids)
        steps += 1
        assert steps < 3000
    for rid, prompt in prompts.items():
        seq = list(prompt)
        for _ in range(lens[rid]):
            seq.append(next_token(seq))
        assert finished[rid] == seq[len(prompt):], rid
    assert engine.scheduler.block_manager.get_num_free_gpu_blocks() == cc.num_gpu_blocks
    assert (getattr(engine.scheduler, "num_preemptions", 0) > 0) == (blocks == 2)
    with pytest.raises(ValueError):
        EngineArgs(model="llama-tiny", overlap_prompt_decode=True).create_engine_configs()       # needs pipeline_steps


def test_overlap_engine_shares_blocks_that_decode_steps_filled(tmp_path, monkeypatch):
    """One GPU, one cache: with the prefix cache on, a later prompt that starts with an earlier request's prompt AND
    answer (the next turn of a conversation) finds the blocks the decode steps filled and does not compute them
    again -- as in the co-located engine, unlike two pools, where a decode step's keys and values exist on the token
    pool only.  The stand-in worker takes a cached prompt's token from what its cache holds, so a block reported as
    computed before it was written, or shared with other content, gives a wrong token."""
    from split_fakes import FakeWorker
    from splitwise_amd.executor.overlap_executor import OverlapExecutor
    monkeypatch.setenv("SPLIT_FAKE_DIR", str(tmp_path))
    monkeypatch.setenv("SPLIT_FAKE_BLOCKS", "48")
    args = EngineArgs(model="llama-tiny", max_num_seqs=6, max_model_len=256, max_num_batched_tokens=512,
                      swap_space=0.001, pipeline_steps=True, overlap_prompt_decode=True, enable_prefix_caching=True)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    executor = OverlapExecutor(mc, cc, pc, sc, dc, worker_cls=FakeWorker)
    engine = LLMEngine(mc, cc, pc, sc, dc, executor=executor)
    assert engine.scheduler.block_manager.promote_decode_blocks

    def generate(requests):
        for rid, (prompt, n) in requests.items():
            engine.add_request(rid, None, SamplingParams(temperature=0.0, max_tokens=n, ignore_eos=True),
                               prompt_token_ids=prompt)
        done, calls = {}, 0
        while engine.has_unfinished_requests():
            calls += 1
            assert calls < 3000
            for out in engine.step():
                if out.finished:
                    done[out.request_id] = list(out.outputs[0].token_ids)
        return done

    def expected(prompt, n):
        seq = list(prompt)
        for _ in range(n):
            seq.append(next_token(seq))
        return seq[len(prompt):]

    first = {"a": ([(7 * j + 3) % VOCAB for j in range(21)], 60), "b": ([(5 * j + 1) % VOCAB for j in range(33)], 45)}
    out1 = generate(first)
    assert all(out1[rid] == expected(*first[rid]) for rid in first)
    worker = executor.driver_worker
    before = worker.cached_prompt_tokens
    # the next turn: prompt + answer + a few new tokens (and one that parts ways in the middle of the answer)
    second = {"a2": (first["a"][0] + out1["a"] + [9, 8, 7], 20),
              "b2": (first["b"][0] + out1["b"][:30] + [4, 4, 4, 4], 20)}
    out2 = generate(second)
    assert all(out2[rid] == expected(*second[rid]) for rid in second)
    # a's 81 tokens fill 5 blocks (4 reported as computed: never the last full one), b's first 63 tokens 3
    assert worker.cached_prompt_tokens - before >= (4 + 3) * 16
    assert engine.scheduler.block_manager.get_num_free_gpu_blocks() == cc.num_gpu_blocks
