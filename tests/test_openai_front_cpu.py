"""The OpenAI-style front end with a tokenizer, on CPU: the real engine, scheduler and async loop over
the deterministic stand-in executor of test_async_engine_cpu (next token = a fixed function of the
sequence), so every expected text is computable.  Mirrors what the reference's
tests/entrypoints/test_openai_server.py asks of its server (single / batch completions, streaming
equals non-streaming, chat, logprobs, errors) without a hub model."""
import json

import pytest
import torch

import conftest  # noqa: F401
from split_fakes import VOCAB, next_token
from test_async_engine_cpu import FakeExecutor
from tiny_tokenizer import build_tokenizer
from splitwise_amd import EngineArgs, LLMEngine, SamplingParams
from splitwise_amd.engine.async_llm_engine import AsyncLLMEngine
from splitwise_amd.transformers_utils.tokenizer import detokenize_incrementally

TOK = build_tokenizer()
assert len(TOK) >= VOCAB


def _engine(log_stats=False, max_model_len=256) -> LLMEngine:
    args = EngineArgs(model="llama-tiny", max_num_seqs=8, max_model_len=max_model_len, max_num_batched_tokens=256,
                      swap_space=0.001)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    cc.num_gpu_blocks, cc.num_cpu_blocks = 128, 8
    return LLMEngine(mc, cc, pc, sc, dc, executor=FakeExecutor(), tokenizer=TOK, log_stats=log_stats)


def _expected_ids(prompt_ids, n):
    seq = list(prompt_ids)
    for _ in range(n):
        seq.append(next_token(seq))
    return seq[len(prompt_ids):]


def _text(prompt_ids, out_ids):
    """Text of the output ids as a stream reader sees it: like one-shot decoding, except that a trailing
    piece which ends inside a multi-byte character is never emitted (the equivalence with
    ``decode`` is what test_incremental_detokenisation_equals_one_shot_decoding checks)."""
    tokens, po, ro, text = None, 0, 0, ""
    for i in range(len(out_ids)):
        new, piece, po, ro = detokenize_incrementally(TOK, list(prompt_ids) + list(out_ids[:i + 1]), tokens, po, ro,
                                                      skip_special_tokens=True)
        tokens = new if tokens is None else tokens + new
        text += piece
    return text


def _client(engine=None, **kw):
    from fastapi.testclient import TestClient
    from splitwise_amd.entrypoints.openai import api_server
    return TestClient(api_server.build_app(AsyncLLMEngine(engine=engine or _engine()), "llama-tiny", **kw))


def _sse(resp):
    lines = [ln for ln in resp.iter_lines() if ln.startswith("data: ")]
    assert lines[-1] == "data: [DONE]"
    return [json.loads(ln[6:]) for ln in lines[:-1]]


# ------------------------------------------------------------------ detokeniser
def test_incremental_detokenisation_equals_one_shot_decoding():
    import random
    random.seed(0)
    prompt = TOK("the quick brown").input_ids
    for _ in range(100):
        out = [random.randrange(3, len(TOK)) for _ in range(random.randint(1, 40))]
        tokens, po, ro, text = None, 0, 0, ""
        for i in range(len(out)):
            new, piece, po, ro = detokenize_incrementally(TOK, prompt + out[:i + 1], tokens, po, ro,
                                                          skip_special_tokens=True)
            tokens = new if tokens is None else tokens + new
            text += piece
        full = TOK.decode(out, skip_special_tokens=True)
        assert full.startswith(text)
        if not full.endswith("�"):
            assert text == full


def test_multibyte_character_is_held_back_until_complete():
    ids = TOK("🙂").input_ids
    assert len(ids) > 1                                   # the emoji spans several byte-level tokens
    prompt = TOK("hello").input_ids
    tokens, po, ro, pieces = None, 0, 0, []
    for i in range(len(ids)):
        new, piece, po, ro = detokenize_incrementally(TOK, prompt + ids[:i + 1], tokens, po, ro)
        tokens = new if tokens is None else tokens + new
        pieces.append(piece)
    assert "".join(pieces) == "🙂" and all(p == "" for p in pieces[:-1])


def test_special_tokens_are_skipped_or_kept():
    ids = TOK("hello").input_ids + [TOK.eos_token_id]
    for skip, want in ((True, ""), (False, "</s>")):
        tokens, po, ro = None, 0, 0
        for i in range(len(ids)):
            new, piece, po, ro = detokenize_incrementally(TOK, ids[:i + 1], tokens, po, ro, skip_special_tokens=skip)
            tokens = new if tokens is None else tokens + new
        assert piece == want


# ------------------------------------------------------------------ completions
def test_text_completion_usage_echo_and_n():
    prompt = "the quick brown fox"
    ids = TOK(prompt).input_ids
    want_ids = _expected_ids(ids, 6)
    want_text = _text(ids, want_ids)
    with _client() as client:
        r = client.post("/v1/completions", json={"model": "llama-tiny", "prompt": prompt, "max_tokens": 6,
                                                 "temperature": 0.0, "ignore_eos": True})
        body = r.json()
        assert r.status_code == 200, body
        assert body["choices"][0]["text"] == want_text and body["choices"][0]["token_ids"] == want_ids
        assert body["usage"] == {"prompt_tokens": len(ids), "completion_tokens": 6, "total_tokens": len(ids) + 6}
        assert body["id"].startswith("cmpl-") and body["model"] == "llama-tiny"
        r = client.post("/v1/completions", json={"prompt": [prompt, "hello world"], "max_tokens": 4, "n": 2,
                                                 "temperature": 0.0, "ignore_eos": True, "echo": True})
        body = r.json()
        assert [c["index"] for c in body["choices"]] == [0, 1, 2, 3]
        assert body["choices"][0]["text"] == body["choices"][1]["text"] == prompt + _text(ids, _expected_ids(ids, 4))
        hello = TOK("hello world").input_ids
        assert body["choices"][3]["text"] == "hello world" + _text(hello, _expected_ids(hello, 4))
        assert body["usage"]["prompt_tokens"] == 2 * (len(ids) + len(hello))


def test_stop_string_ends_and_is_cut():
    prompt = "hello world"
    ids = TOK(prompt).input_ids
    full = _text(ids, _expected_ids(ids, 24))
    stop = full[len(full) // 2:len(full) // 2 + 3]
    cut = full.index(stop)
    with _client() as client:
        r = client.post("/v1/completions", json={"prompt": prompt, "max_tokens": 24, "temperature": 0.0,
                                                 "ignore_eos": True, "stop": [stop]})
        c = r.json()["choices"][0]
        assert c["finish_reason"] == "stop" and c["text"] == full[:cut]
        r = client.post("/v1/completions", json={"prompt": prompt, "max_tokens": 24, "temperature": 0.0,
                                                 "ignore_eos": True, "stop": stop, "include_stop_str_in_output": True})
        assert r.json()["choices"][0]["text"] == full[:cut + len(stop)]


def test_streaming_equals_plain_answer():
    prompt = "prompt and token machines"
    req = {"prompt": prompt, "max_tokens": 9, "temperature": 0.0, "ignore_eos": True, "logprobs": 1}
    with _client() as client:
        plain = client.post("/v1/completions", json=req).json()
        with client.stream("POST", "/v1/completions", json=dict(req, stream=True)) as resp:
            events = _sse(resp)
    chunks = [e for e in events if e["choices"]]
    assert "".join(c["choices"][0]["text"] for c in chunks) == plain["choices"][0]["text"]
    assert [c["choices"][0]["finish_reason"] for c in chunks][-1] == "length"
    assert all(c["choices"][0]["finish_reason"] is None for c in chunks[:-1])
    assert events[-1]["choices"] == [] and events[-1]["usage"] == plain["usage"]
    # log-probabilities: one entry per token, offsets continue across chunks
    lp = plain["choices"][0]["logprobs"]
    assert len(lp["tokens"]) == len(lp["token_logprobs"]) == len(lp["text_offset"]) == len(lp["top_logprobs"]) == 9
    assert "".join(lp["tokens"]) == plain["choices"][0]["text"]
    streamed_offsets = [o for c in chunks for o in c["choices"][0]["logprobs"]["text_offset"]]
    assert streamed_offsets == lp["text_offset"]


def test_request_errors():
    with _client(_engine(max_model_len=32)) as client:
        r = client.post("/v1/completions", json={"prompt": "hello world", "max_tokens": 64})
        assert r.status_code == 400 and "maximum context length is 32 tokens" in r.json()["message"]
        for bad in ({"presence_penalty": 0.5}, {"use_beam_search": True}, {"logit_bias": {"1": 2.0}}, {"best_of": 3},
                    {"suffix": "x"}, {"top_p": 3.0}, {"n": 0}):
            r = client.post("/v1/completions", json=dict({"prompt": "hello"}, **bad))
            assert r.status_code == 400 and r.json()["object"] == "error", bad
        r = client.post("/v1/completions", json={"model": "gpt-5", "prompt": "hello"})
        assert r.status_code == 404 and r.json()["type"] == "NotFoundError"
        assert client.post("/v1/completions", content=b"{not json").status_code == 400
        assert client.post("/v1/completions", json=[1, 2]).status_code == 400
        assert client.post("/v1/completions", json={"prompt": []}).status_code == 400
        # the server keeps serving afterwards
        assert client.post("/v1/completions", json={"prompt": "hello", "max_tokens": 2}).status_code == 200


def test_api_key():
    with _client(api_key="sesame") as client:
        assert client.get("/health").status_code == 200
        assert client.get("/v1/models").status_code == 401
        assert client.get("/v1/models", headers={"Authorization": "Bearer wrong"}).status_code == 401
        assert client.get("/v1/models", headers={"Authorization": "Bearer sesame"}).status_code == 200


# ------------------------------------------------------------------------- chat
def test_chat_completion_plain_and_streamed():
    messages = [{"role": "system", "content": "be brief"}, {"role": "user", "content": "what is the answer?"}]
    rendered = TOK.apply_chat_template(conversation=messages, tokenize=False, add_generation_prompt=True)
    ids = TOK(rendered).input_ids
    want = _text(ids, _expected_ids(ids, 7))
    req = {"model": "llama-tiny", "messages": messages, "max_tokens": 7, "temperature": 0.0, "ignore_eos": True}
    with _client() as client:
        r = client.post("/v1/chat/completions", json=req)
        body = r.json()
        assert r.status_code == 200, body
        assert body["object"] == "chat.completion" and body["id"].startswith("chatcmpl-")
        assert body["choices"][0]["message"] == {"role": "assistant", "content": want}
        assert body["choices"][0]["finish_reason"] == "length"
        assert body["usage"] == {"prompt_tokens": len(ids), "completion_tokens": 7, "total_tokens": len(ids) + 7}
        with client.stream("POST", "/v1/chat/completions", json=dict(req, stream=True)) as resp:
            events = _sse(resp)
        assert events[0]["object"] == "chat.completion.chunk"
        assert events[0]["choices"][0]["delta"] == {"role": "assistant"}
        content = "".join(e["choices"][0]["delta"].get("content", "") for e in events[1:])
        assert content == want
        assert events[-1]["choices"][0]["finish_reason"] == "length" and events[-1]["choices"][0]["delta"] == {}
        assert events[-1]["usage"]["completion_tokens"] == 7
        # default max_tokens: whatever the context has left
        r = client.post("/v1/chat/completions", json={"messages": messages, "temperature": 0.0, "ignore_eos": True})
        assert r.json()["usage"]["total_tokens"] == 256
        # two choices; logprobs with top_logprobs
        r = client.post("/v1/chat/completions", json=dict(req, n=2, logprobs=True, top_logprobs=1))
        body = r.json()
        assert [c["index"] for c in body["choices"]] == [0, 1]
        assert len(body["choices"][0]["logprobs"]["tokens"]) == 7
        assert client.post("/v1/chat/completions", json={"messages": []}).status_code == 400
        assert client.post("/v1/chat/completions", json={"messages": [{"role": "user"}]}).status_code == 400


def test_chat_template_from_the_command_line_and_response_role():
    template = "{% for m in messages %}[{{ m['role'] }}] {{ m['content'] }}\\n{% endfor %}[bot]"
    with _client(chat_template=template, response_role="bot") as client:
        messages = [{"role": "user", "content": "hello"}]
        r = client.post("/v1/chat/completions", json={"messages": messages, "max_tokens": 3, "temperature": 0.0,
                                                      "ignore_eos": True})
        ids = TOK("[user] hello\n[bot]").input_ids
        body = r.json()
        assert body["usage"]["prompt_tokens"] == len(ids)
        assert body["choices"][0]["message"] == {"role": "bot", "content": _text(ids, _expected_ids(ids, 3))}
    TOK.chat_template = __import__("tiny_tokenizer").CHAT_TEMPLATE     # the tokenizer object is shared by the module


def test_chat_needs_a_template():
    args = EngineArgs(model="llama-tiny", max_num_seqs=4, max_model_len=64, max_num_batched_tokens=64, swap_space=0.001)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    cc.num_gpu_blocks, cc.num_cpu_blocks = 16, 8
    with _client(LLMEngine(mc, cc, pc, sc, dc, executor=FakeExecutor())) as client:
        r = client.post("/v1/chat/completions", json={"messages": [{"role": "user", "content": "hi"}]})
        assert r.status_code == 400 and "chat template" in r.json()["message"]
        r = client.post("/v1/completions", json={"prompt": "text without a tokenizer"})
        assert r.status_code == 400 and "token ids" in r.json()["message"]


# ---------------------------------------------------------------------- metrics
def test_prometheus_series_follow_the_requests():
    with _client(_engine(log_stats=True)) as client:
        for _ in range(3):
            assert client.post("/v1/completions", json={"prompt": "hello world", "max_tokens": 5, "temperature": 0.0,
                                                        "ignore_eos": True}).status_code == 200
        text = client.get("/metrics/").text
    label = '{model_name="llama-tiny"}'

    def value(series):
        for line in text.splitlines():
            if line.startswith(series + " ") or line.startswith(series + "{"):
                if series + label in line or "{" not in line:
                    return float(line.rsplit(" ", 1)[1])
        raise AssertionError(f"{series} not exported:\n{text[:2000]}")

    n_prompt = len(TOK("hello world").input_ids)
    assert value("vllm:prompt_tokens_total") == 3 * n_prompt
    assert value("vllm:generation_tokens_total") == 3 * 4            # a prompt step yields a token that is not counted here
    assert value("vllm:time_to_first_token_seconds_count") == 3
    assert value("vllm:time_per_output_token_seconds_count") == 3 * 4
    assert value("vllm:e2e_request_latency_seconds_count") == 3
    assert value("vllm:num_requests_running") == 0 and value("vllm:num_requests_waiting") == 0
    for series in ("vllm:gpu_cache_usage_perc", "vllm:cpu_cache_usage_perc", "vllm:num_requests_swapped",
                   "vllm:avg_prompt_throughput_toks_per_s", "vllm:avg_generation_throughput_toks_per_s"):
        value(series)
    assert 'vllm:cache_config_info{' in text and 'block_size="16"' in text
    assert 'vllm:time_to_first_token_seconds_bucket{le="0.25",model_name="llama-tiny"}' in text


def test_interval_throughput_uses_the_engine_clock():
    from splitwise_amd.engine.metrics import StatLogger, Stats
    log = StatLogger(local_interval=1.0, labels={"model_name": "m"})
    t0 = log.last_local_log

    def stats(now, p, g):
        return Stats(now=now, num_running=1, num_swapped=0, num_waiting=0, gpu_cache_usage=0.5, cpu_cache_usage=0.0,
                     num_prompt_tokens=p, num_generation_tokens=g)

    log.log(stats(t0 + 0.5, 100, 0))
    assert log.num_prompt_tokens == [100]
    log.log(stats(t0 + 2.0, 0, 50))                       # 2 s of engine time: interval over, counters reset
    assert log.num_prompt_tokens == [] and log.last_local_log == t0 + 2.0
    assert log.metrics.gauge_avg_prompt_throughput.labels(model_name="m")._value.get() == pytest.approx(50.0)
    assert log.metrics.gauge_avg_generation_throughput.labels(model_name="m")._value.get() == pytest.approx(25.0)


# ---------------------------------------------------------------------- sampler
def test_sampler_returns_the_requested_top_logprobs():
    from splitwise_amd.model_executor.layers.sampler import Sampler, SamplingMetadata
    torch.manual_seed(0)
    logits = torch.randn(3, 50)
    groups = [([0], SamplingParams(temperature=0.0, logprobs=3)), ([1], SamplingParams(temperature=0.0)),
              ([2], SamplingParams(temperature=0.0, logprobs=0))]
    out = Sampler(50).sample_from_logits(logits, SamplingMetadata(groups, None, None, torch.arange(3)))
    ref = torch.log_softmax(logits, dim=-1)
    first = out.outputs[0].samples[0]
    assert first.output_token == int(ref[0].argmax())
    assert sorted(first.logprobs) == sorted(ref[0].topk(3).indices.tolist())
    for tok, entry in first.logprobs.items():
        assert entry.logprob == pytest.approx(float(ref[0, tok]), abs=1e-6)
    for row in (1, 2):                                    # not asked for / asked for none: the sampled token only
        s = out.outputs[row].samples[0]
        assert list(s.logprobs) == [s.output_token]
        assert s.logprobs[s.output_token].logprob == pytest.approx(float(ref[row].max()), abs=1e-6)


def test_sampler_greedy_fast_path_reports_only_the_sampled_token():
    from splitwise_amd.model_executor.layers.sampler import Sampler, SamplingMetadata
    torch.manual_seed(1)
    logits = torch.randn(2, 50)
    groups = [([0], SamplingParams(temperature=0.0)), ([1], SamplingParams(temperature=0.0))]
    out = Sampler(50).sample_from_logits(logits, SamplingMetadata(groups, None, None, torch.arange(2)))
    ref = torch.log_softmax(logits, dim=-1)
    for row in (0, 1):
        s = out.outputs[row].samples[0]
        assert s.output_token == int(ref[row].argmax()) and list(s.logprobs) == [s.output_token]
        assert s.logprobs[s.output_token].logprob == pytest.approx(float(ref[row].max()), abs=1e-5)
