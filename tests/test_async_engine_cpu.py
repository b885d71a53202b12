"""AsyncLLMEngine and the two HTTP fronts on CPU: the real engine and scheduler over a
deterministic stand-in executor (next token = a fixed function of the whole sequence)."""
import asyncio
import json

import pytest

import conftest  # noqa: F401
from split_fakes import VOCAB, next_token
from splitwise_amd import EngineArgs, LLMEngine, SamplingParams
from splitwise_amd.engine.async_llm_engine import AsyncLLMEngine
from splitwise_amd.executor.executor_base import ExecutorBase
from splitwise_amd.sequence import Logprob, SamplerOutput, SequenceGroupOutput, SequenceOutput


class FakeExecutor(ExecutorBase):
    def execute_model(self, seq_group_metadata_list, blocks_to_swap_in, blocks_to_swap_out, blocks_to_copy,
                      blocks_to_nw=None):
        outs = []
        for meta in seq_group_metadata_list:
            (seq_id, data), = meta.seq_data.items()
            tok = next_token(data.get_token_ids())
            outs.append(SequenceGroupOutput([SequenceOutput(seq_id, tok, {tok: Logprob(0.0)})]))
        return SamplerOutput(outs)

    def check_health(self) -> None:
        pass


def _engine() -> LLMEngine:
    args = EngineArgs(model="llama-tiny", max_num_seqs=4, max_model_len=256, max_num_batched_tokens=256,
                      swap_space=0.001)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    cc.num_gpu_blocks, cc.num_cpu_blocks = 64, 8
    return LLMEngine(mc, cc, pc, sc, dc, executor=FakeExecutor())


def _expected(prompt, n):
    seq = list(prompt)
    for _ in range(n):
        seq.append(next_token(seq))
    return seq[len(prompt):]


def test_generate_streams_every_output_and_requests_interleave():
    async def scenario():
        engine = AsyncLLMEngine(engine=_engine())
        prompts = {f"r{i}": [(7 * i + j) % VOCAB for j in range(3 + 5 * i)] for i in range(6)}

        async def one(rid, prompt, n):
            seen = []
            async for out in engine.generate(None, SamplingParams(temperature=0.0, max_tokens=n, ignore_eos=True), rid,
                                             prompt_token_ids=prompt):
                seen.append(list(out.outputs[0].token_ids))
            return rid, seen

        results = await asyncio.gather(*[one(rid, p, 4 + i) for i, (rid, p) in enumerate(prompts.items())])
        for i, (rid, seen) in enumerate(results):
            want = _expected(prompts[rid], 4 + i)
            assert seen[-1] == want
            assert [len(s) for s in seen] == list(range(1, len(want) + 1))      # one new token per output
        await engine.check_health()
        engine.background_loop.cancel()
    asyncio.run(scenario())


def test_abort_ends_the_stream_and_frees_the_request():
    async def scenario():
        engine = AsyncLLMEngine(engine=_engine())
        got = []

        async def consume():
            async for out in engine.generate(None, SamplingParams(temperature=0.0, max_tokens=200, ignore_eos=True),
                                             "long", prompt_token_ids=[1, 2, 3]):
                got.append(len(out.outputs[0].token_ids))
                if len(got) == 3:
                    await engine.abort("long")
        await asyncio.wait_for(consume(), timeout=30)
        assert 3 <= len(got) < 200
        for _ in range(50):
            if not engine.engine.has_unfinished_requests():
                break
            await asyncio.sleep(0.01)
        assert not engine.engine.has_unfinished_requests()
        engine.background_loop.cancel()
    asyncio.run(scenario())


def test_bad_requests_fail_their_own_stream_only():
    async def scenario():
        engine = AsyncLLMEngine(engine=_engine())
        with pytest.raises(ValueError, match="tokenizer"):
            async for _ in engine.generate("text prompt without a tokenizer", SamplingParams(max_tokens=2), "bad"):
                pass
        outs = [o async for o in engine.generate(None, SamplingParams(temperature=0.0, max_tokens=2), "good",
                                                 prompt_token_ids=[5, 6])]
        assert outs[-1].finished
        engine.background_loop.cancel()
    asyncio.run(scenario())


def test_http_generate_and_openai_routes():
    from fastapi.testclient import TestClient

    from splitwise_amd.entrypoints import api_server
    from splitwise_amd.entrypoints.openai import api_server as openai_server
    engine = AsyncLLMEngine(engine=_engine())
    prompt = [3, 1, 4, 1, 5, 9, 2, 6]
    with TestClient(api_server.build_app(engine)) as client:
        assert client.get("/health").status_code == 200
        r = client.post("/generate", json={"prompt_token_ids": prompt, "max_tokens": 5, "temperature": 0.0,
                                           "ignore_eos": True})
        assert r.status_code == 200
        body = r.json()
        assert body["token_ids"] == [_expected(prompt, 5)] and len(body["timestamps"]) == 5
        assert client.post("/generate", json={"max_tokens": 5}).status_code == 400
        assert client.post("/generate", json={"prompt_token_ids": prompt, "top_p": 7.0}).status_code == 400
        with client.stream("POST", "/generate", json={"prompt_token_ids": prompt, "max_tokens": 3, "temperature": 0.0,
                                                      "ignore_eos": True, "stream": True}) as resp:
            chunks = [json.loads(c) for c in b"".join(resp.iter_bytes()).split(b"\0") if c]
        assert [len(c["token_ids"][0]) for c in chunks] == [1, 2, 3]
    engine = AsyncLLMEngine(engine=_engine())
    with TestClient(openai_server.build_app(engine, "llama-tiny")) as client:
        metrics = client.get("/metrics/")
        assert metrics.status_code == 200 and "vllm:num_requests_running" in metrics.text
        assert client.get("/v1/models").json()["data"][0]["id"] == "llama-tiny"
        r = client.post("/v1/completions", json={"model": "llama-tiny", "prompt": [prompt, prompt[:3]], "max_tokens": 4,
                                                 "temperature": 0.0, "ignore_eos": True})
        body = r.json()
        assert r.status_code == 200 and body["object"] == "text_completion"
        assert [c["token_ids"] for c in body["choices"]] == [_expected(prompt, 4), _expected(prompt[:3], 4)]
        assert body["usage"] == {"prompt_tokens": 11, "completion_tokens": 8, "total_tokens": 19}
        assert body["choices"][0]["finish_reason"] == "length"
        assert client.post("/v1/completions", json={"model": "other", "prompt": prompt}).status_code == 404
        assert client.post("/v1/completions", json={"prompt": {"not": "valid"}}).status_code == 400
        with client.stream("POST", "/v1/completions", json={"prompt": prompt, "max_tokens": 3, "temperature": 0.0,
                                                            "ignore_eos": True, "stream": True}) as resp:
            lines = [ln for ln in resp.iter_lines() if ln.startswith("data: ")]
        assert lines[-1] == "data: [DONE]" and len(lines) == 5          # 3 tokens, the usage count, the end mark
        assert json.loads(lines[2][6:])["choices"][0]["finish_reason"] == "length"
        assert json.loads(lines[3][6:])["usage"] == {"prompt_tokens": 8, "completion_tokens": 3, "total_tokens": 11}
        assert client.post("/v1/chat/completions", json={"messages": [{"role": "user", "content": "hi"}]}).status_code == 400


def test_engine_stats_use_one_clock():
    """TTFT / TPOT / e2e / time in queue are differences against the request's monotonic arrival time:
    they must come out as fractions of a second, not as the wall clock minus a monotonic reading."""
    import conftest  # noqa: F401
    from test_pipelined_engine_cpu import FakeExecutor
    from splitwise_amd import EngineArgs, LLMEngine, SamplingParams
    args = EngineArgs(model="llama-tiny", max_num_seqs=4, max_model_len=256, max_num_batched_tokens=256, swap_space=0.001)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    cc.num_gpu_blocks, cc.num_cpu_blocks = 16, 8
    engine = LLMEngine(mc, cc, pc, sc, dc, executor=FakeExecutor())
    seen = []
    engine._get_stats_orig = engine._get_stats

    def spy(*a, **k):
        stats = engine._get_stats_orig(*a, **k)
        seen.append(stats)
        return stats

    engine._get_stats = spy

    class _Log:
        def log(self, stats):
            pass

    engine.stat_logger = _Log()
    engine.add_request("r", None, SamplingParams(temperature=0.0, max_tokens=3, ignore_eos=True), prompt_token_ids=[1, 2, 3])
    group = engine.scheduler.waiting[0]
    while engine.has_unfinished_requests():
        engine.step()
    assert seen
    for stats in seen:
        for v in stats.time_to_first_tokens + stats.time_per_output_tokens + stats.time_e2e_requests:
            assert 0.0 <= v < 5.0, v
    assert 0.0 <= group.metrics.time_in_queue < 5.0
    assert group.metrics.first_token_time > 1.0e9            # wall clock, as in the reference
