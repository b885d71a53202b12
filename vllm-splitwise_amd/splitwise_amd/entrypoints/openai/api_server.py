"""OpenAI-style routes (vllm/entrypoints/openai/api_server.py:150-196): ``/metrics``, ``/health``,
``/v1/models``, ``/v1/completions`` (plain and server-sent-event streaming) and
``/v1/chat/completions``.  Prompts may be strings (with a tokenizer on the engine), one list of
token ids, or a list of such lists, as the OpenAI completions API allows.  Chat needs a
tokenizer with a chat template and answers 400 without one.
"""
import argparse
import json
import time
import uuid
from typing import AsyncGenerator, List, Optional, Union

from fastapi import FastAPI, Request
from fastapi.responses import JSONResponse, Response, StreamingResponse

from ...engine.arg_utils import EngineArgs
from ...engine.async_llm_engine import AsyncLLMEngine
from ...sampling_params import SamplingParams


def _error(message: str, status: int = 400, kind: str = "invalid_request_error") -> JSONResponse:
    return JSONResponse({"object": "error", "message": message, "type": kind, "code": status}, status_code=status)


def _prompts(raw) -> List[Union[str, List[int]]]:
    if isinstance(raw, str):
        return [raw]
    if isinstance(raw, list) and raw and all(isinstance(t, int) for t in raw):
        return [raw]
    if isinstance(raw, list) and raw and all(isinstance(p, (str, list)) for p in raw):
        return list(raw)
    raise ValueError("prompt must be a string, a list of token ids, or a list of those")


def _sampling_params(body: dict) -> SamplingParams:
    return SamplingParams(n=body.get("n", 1), best_of=body.get("best_of"), temperature=body.get("temperature", 1.0),
                          top_p=body.get("top_p", 1.0), top_k=body.get("top_k", -1), stop=body.get("stop"),
                          stop_token_ids=body.get("stop_token_ids"), ignore_eos=body.get("ignore_eos", False),
                          max_tokens=body.get("max_tokens", 16), logprobs=body.get("logprobs"),
                          seed=body.get("seed"))


def build_app(engine: AsyncLLMEngine, served_model: str) -> FastAPI:
    app = FastAPI()
    try:                                            # /metrics (openai/api_server.py:150-152)
        from prometheus_client import make_asgi_app
        from ...engine.metrics import _metrics
        _metrics()                                  # registers the vllm:* series even before the first step
        app.mount("/metrics", make_asgi_app())
    except ImportError:                             # pragma: no cover - the front works without it
        pass

    @app.get("/health")
    async def health() -> Response:
        await engine.check_health()
        return Response(status_code=200)

    @app.get("/v1/models")
    async def models() -> JSONResponse:
        cfg = await engine.get_model_config()
        return JSONResponse({"object": "list", "data": [{"id": served_model, "object": "model",
                                                         "created": int(time.time()), "owned_by": "splitwise_amd",
                                                         "max_model_len": cfg.max_model_len}]})

    @app.post("/v1/completions")
    async def completions(request: Request):
        body = await request.json()
        if body.get("model") not in (None, served_model):
            return _error(f"The model `{body.get('model')}` does not exist.", 404, "NotFoundError")
        try:
            prompts = _prompts(body.get("prompt"))
            params = _sampling_params(body)
        except (TypeError, ValueError, NotImplementedError) as e:
            return _error(str(e))
        request_id = f"cmpl-{uuid.uuid4().hex}"
        created = int(time.time())
        gens = []
        for i, p in enumerate(prompts):
            text, ids = (p, None) if isinstance(p, str) else (None, p)
            gens.append(engine.generate(text, params, f"{request_id}-{i}", prompt_token_ids=ids))

        def chunk(index: int, out, delta_from: int, final: bool) -> dict:
            o = out.outputs[0]
            return {"id": request_id, "object": "text_completion", "created": created, "model": served_model,
                    "choices": [{"index": index, "text": o.text[delta_from:] if delta_from >= 0 else o.text,
                                 "token_ids": o.token_ids, "logprobs": None,
                                 "finish_reason": o.finish_reason if final else None}]}

        if body.get("stream", False):
            async def sse() -> AsyncGenerator[str, None]:
                try:
                    for i, gen in enumerate(gens):
                        seen = 0
                        async for out in gen:
                            yield f"data: {json.dumps(chunk(i, out, seen, out.finished))}\n\n"
                            seen = len(out.outputs[0].text)
                except (ValueError, NotImplementedError) as e:
                    yield f"data: {json.dumps({'error': str(e)})}\n\n"
                yield "data: [DONE]\n\n"
            return StreamingResponse(sse(), media_type="text/event-stream")

        choices, prompt_tokens, completion_tokens = [], 0, 0
        try:
            for i, gen in enumerate(gens):
                final = None
                async for out in gen:
                    if await request.is_disconnected():
                        await engine.abort([f"{request_id}-{j}" for j in range(len(gens))])
                        return _error("Client disconnected", 499)
                    final = out
                choices += chunk(i, final, -1, True)["choices"]
                prompt_tokens += len(final.prompt_token_ids)
                completion_tokens += len(final.outputs[0].token_ids)
        except (ValueError, NotImplementedError) as e:
            return _error(str(e))
        return JSONResponse({"id": request_id, "object": "text_completion", "created": created, "model": served_model,
                             "choices": choices,
                             "usage": {"prompt_tokens": prompt_tokens, "completion_tokens": completion_tokens,
                                       "total_tokens": prompt_tokens + completion_tokens}})

    @app.post("/v1/chat/completions")
    async def chat_completions(request: Request):
        body = await request.json()
        tokenizer = engine.engine.tokenizer
        if tokenizer is None or not hasattr(tokenizer, "apply_chat_template"):
            return _error("chat completions need a tokenizer with a chat template; this engine was started "
                          "without one (send token ids to /v1/completions instead)")
        try:
            prompt = tokenizer.apply_chat_template(conversation=body["messages"], tokenize=False,
                                                   add_generation_prompt=body.get("add_generation_prompt", True))
            params = _sampling_params(body)
        except Exception as e:                       # noqa: BLE001 - template errors are the caller's
            return _error(str(e))
        request_id = f"chatcmpl-{uuid.uuid4().hex}"
        final = None
        async for out in engine.generate(prompt, params, request_id):
            final = out
        o = final.outputs[0]
        n_prompt, n_out = len(final.prompt_token_ids), len(o.token_ids)
        return JSONResponse({"id": request_id, "object": "chat.completion", "created": int(time.time()),
                             "model": served_model,
                             "choices": [{"index": 0, "message": {"role": "assistant", "content": o.text},
                                          "finish_reason": o.finish_reason}],
                             "usage": {"prompt_tokens": n_prompt, "completion_tokens": n_out,
                                       "total_tokens": n_prompt + n_out}})

    return app


def main() -> None:
    import uvicorn
    parser = argparse.ArgumentParser()
    parser.add_argument("--host", type=str, default=None)
    parser.add_argument("--port", type=int, default=8000)
    parser.add_argument("--served-model-name", type=str, default=None)
    parser = EngineArgs.add_cli_args(parser)
    args = parser.parse_args()
    engine = AsyncLLMEngine.from_engine_args(EngineArgs.from_cli_args(args))
    uvicorn.run(build_app(engine, args.served_model_name or args.model), host=args.host, port=args.port,
                log_level="info", timeout_keep_alive=5)


if __name__ == "__main__":
    main()
