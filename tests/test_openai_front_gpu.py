"""The HTTP front end over the REAL engine on the GPU (scheduler -> worker -> HIP kernels, HIP-graph
decode): text in, text out through a local tokenizer; the generated tokens and the reported
log-probabilities are judged by the whole-model CPU oracle, streaming against the plain answer,
and the Prometheus series against what was served."""
import json

import pytest
import torch

from oracle import model as omodel
from tiny_tokenizer import build_tokenizer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["plain", "overlap"])
def served(request):
    """The front over the plain engine, and over the engine that runs prompt steps beside its decode steps."""
    from fastapi.testclient import TestClient
    from splitwise_amd import EngineArgs, LLMEngine
    from splitwise_amd.engine.async_llm_engine import AsyncLLMEngine
    from splitwise_amd.entrypoints.openai import api_server
    tok = build_tokenizer(512)                      # llama-tiny has 512 embedding rows
    assert len(tok) == 512
    overlap = request.param == "overlap"
    args = EngineArgs(model="llama-tiny", max_num_seqs=8, max_model_len=512, swap_space=0.01,
                      num_gpu_blocks_override=256, pipeline_steps=overlap, overlap_prompt_decode=overlap)
    engine = LLMEngine.from_engine_args(args, tokenizer=tok, log_stats=True)
    runner = engine.model_executor.driver_worker.model_runner
    sd = {k: v.detach().cpu() for k, v in runner.model.state_dict().items()}
    with TestClient(api_server.build_app(AsyncLLMEngine(engine=engine), "llama-tiny")) as client:
        yield client, tok, sd, runner.model_config.hf_config
    del engine
    torch.cuda.empty_cache()


def _sse(resp):
    lines = [ln for ln in resp.iter_lines() if ln.startswith("data: ")]
    assert lines[-1] == "data: [DONE]"
    return [json.loads(ln[6:]) for ln in lines[:-1]]


def test_completion_tokens_and_logprobs_match_the_cpu_model(served):
    client, tok, sd, cfg = served
    prompt = "the quick brown fox jumps over the lazy dog"
    ids = tok(prompt).input_ids
    r = client.post("/v1/completions", json={"model": "llama-tiny", "prompt": prompt, "max_tokens": 12,
                                             "temperature": 0.0, "ignore_eos": True, "logprobs": 2})
    body = r.json()
    assert r.status_code == 200, body
    choice = body["choices"][0]
    gen = choice["token_ids"]
    assert len(gen) == 12 and choice["finish_reason"] == "length"
    assert body["usage"] == {"prompt_tokens": len(ids), "completion_tokens": 12, "total_tokens": len(ids) + 12}
    ref = omodel.logits(cfg["architecture"], sd, cfg, ids + gen)
    ref_lp = torch.log_softmax(torch.as_tensor(ref).float(), dim=-1)
    lp = choice["logprobs"]
    assert len(lp["tokens"]) == len(lp["token_logprobs"]) == len(lp["top_logprobs"]) == 12
    for j, t in enumerate(gen):
        row = ref_lp[len(ids) + j - 1]
        assert row[t] >= row.max() - 3e-2                          # greedy within fp16 noise of the fp32 model
        assert abs(lp["token_logprobs"][j] - float(row[t])) < 5e-2
        assert 2 <= len(lp["top_logprobs"][j]) <= 3                # top-2, plus the sampled token if outside
        assert max(lp["top_logprobs"][j].values()) == pytest.approx(float(row.max()), abs=5e-2)
    assert "".join(lp["tokens"]) == choice["text"]


def test_streaming_and_batches_agree_with_the_plain_answer(served):
    client, tok, _, _ = served
    req = {"prompt": ["hello world", "prompt and token machines hand the key value cache over"], "max_tokens": 10,
           "temperature": 0.0, "ignore_eos": True}
    plain = client.post("/v1/completions", json=req).json()
    with client.stream("POST", "/v1/completions", json=dict(req, stream=True)) as resp:
        events = _sse(resp)
    texts, ids = {0: "", 1: ""}, {0: [], 1: []}
    for e in events:
        for c in e["choices"]:
            texts[c["index"]] += c["text"]
            ids[c["index"]] += c["token_ids"]
    for i in (0, 1):
        assert texts[i] == plain["choices"][i]["text"] and ids[i] == plain["choices"][i]["token_ids"]
    assert events[-1]["usage"] == plain["usage"]
    # a seeded random request is reproducible, and its two samples differ from each other
    seeded = {"prompt": "hello world", "max_tokens": 12, "temperature": 1.0, "seed": 7, "n": 2, "ignore_eos": True}
    a, b = client.post("/v1/completions", json=seeded).json(), client.post("/v1/completions", json=seeded).json()
    assert [c["token_ids"] for c in a["choices"]] == [c["token_ids"] for c in b["choices"]]
    assert a["choices"][0]["token_ids"] != a["choices"][1]["token_ids"]


def test_chat_and_stop_string(served):
    client, tok, _, _ = served
    messages = [{"role": "user", "content": "what is the answer?"}]
    req = {"messages": messages, "max_tokens": 16, "temperature": 0.0, "ignore_eos": True}
    body = client.post("/v1/chat/completions", json=req).json()
    full = body["choices"][0]["message"]["content"]
    rendered = tok.apply_chat_template(conversation=messages, tokenize=False, add_generation_prompt=True)
    assert body["usage"]["prompt_tokens"] == len(tok(rendered).input_ids)
    with client.stream("POST", "/v1/chat/completions", json=dict(req, stream=True)) as resp:
        events = _sse(resp)
    assert events[0]["choices"][0]["delta"] == {"role": "assistant"}
    assert "".join(e["choices"][0]["delta"].get("content", "") for e in events[1:]) == full
    if len(full) >= 4:
        stop = full[len(full) // 2:len(full) // 2 + 2]
        cut = client.post("/v1/chat/completions", json=dict(req, stop=[stop])).json()["choices"][0]
        assert cut["finish_reason"] == "stop" and cut["message"]["content"] == full[:full.index(stop)]


def test_metrics_count_what_was_served(served):
    client, _, _, _ = served
    before = client.get("/metrics/").text

    def value(text, series):
        for line in text.splitlines():
            if line.startswith(series + "{") and 'model_name="llama-tiny"' in line and "le=" not in line:
                return float(line.rsplit(" ", 1)[1])
        raise AssertionError(series)

    r = client.post("/v1/completions", json={"prompt": "hello world", "max_tokens": 6, "temperature": 0.0,
                                             "ignore_eos": True})
    n_prompt = r.json()["usage"]["prompt_tokens"]
    after = client.get("/metrics/").text
    assert value(after, "vllm:prompt_tokens_total") - value(before, "vllm:prompt_tokens_total") == n_prompt
    assert value(after, "vllm:generation_tokens_total") - value(before, "vllm:generation_tokens_total") == 5
    assert value(after, "vllm:time_to_first_token_seconds_count") - value(before, "vllm:time_to_first_token_seconds_count") == 1
    assert value(after, "vllm:e2e_request_latency_seconds_count") - value(before, "vllm:e2e_request_latency_seconds_count") == 1
    assert 0.0 <= value(after, "vllm:gpu_cache_usage_perc") <= 1.0
