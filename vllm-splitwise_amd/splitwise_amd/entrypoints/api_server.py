"""Minimal HTTP front (vllm/entrypoints/api_server.py:22-118): ``GET /health`` and
``POST /generate``.  A request carries ``prompt`` (needs a tokenizer on the engine) or
``prompt_token_ids``, ``stream``, and SamplingParams fields; the answer lists the generated
text, the token ids and one monotonic timestamp per received output (what the reference's
goodput client computes TTFT / TPOT from).
"""
import argparse
import json
import time
import uuid
from typing import AsyncGenerator

from fastapi import FastAPI, Request
from fastapi.responses import JSONResponse, Response, StreamingResponse

from ..engine.arg_utils import EngineArgs
from ..engine.async_llm_engine import AsyncLLMEngine
from ..sampling_params import SamplingParams

TIMEOUT_KEEP_ALIVE = 5  # seconds


def build_app(engine: AsyncLLMEngine) -> FastAPI:
    app = FastAPI()

    @app.get("/health")
    async def health() -> Response:
        await engine.check_health()
        return Response(status_code=200)

    @app.post("/generate")
    async def generate(request: Request) -> Response:
        body = await request.json()
        prompt = body.pop("prompt", None)
        prompt_token_ids = body.pop("prompt_token_ids", None)
        stream = body.pop("stream", False)
        if prompt is None and prompt_token_ids is None:
            return JSONResponse({"error": "prompt or prompt_token_ids is required"}, status_code=400)
        try:
            params = SamplingParams(**body)
        except (TypeError, ValueError, NotImplementedError) as e:
            return JSONResponse({"error": str(e)}, status_code=400)
        request_id = uuid.uuid4().hex
        results = engine.generate(prompt, params, request_id, prompt_token_ids=prompt_token_ids)

        async def stream_results() -> AsyncGenerator[bytes, None]:
            async for out in results:
                ret = {"text": [(out.prompt or "") + o.text for o in out.outputs],
                       "token_ids": [o.token_ids for o in out.outputs]}
                yield (json.dumps(ret) + "\0").encode("utf-8")

        if stream:
            return StreamingResponse(stream_results())
        final, stamps = None, []
        try:
            async for out in results:
                if await request.is_disconnected():
                    await engine.abort(request_id)
                    return Response(status_code=499)
                final = out
                stamps.append(time.perf_counter())
        except (ValueError, NotImplementedError) as e:
            return JSONResponse({"error": str(e)}, status_code=400)
        assert final is not None
        return JSONResponse({"text": [(final.prompt or "") + o.text for o in final.outputs],
                             "token_ids": [o.token_ids for o in final.outputs], "timestamps": stamps})

    return app


def main() -> None:
    import uvicorn
    parser = argparse.ArgumentParser()
    parser.add_argument("--host", type=str, default=None)
    parser.add_argument("--port", type=int, default=8100)
    parser = EngineArgs.add_cli_args(parser)
    args = parser.parse_args()
    engine = AsyncLLMEngine.from_engine_args(EngineArgs.from_cli_args(args))
    uvicorn.run(build_app(engine), host=args.host, port=args.port, log_level="info",
                timeout_keep_alive=TIMEOUT_KEEP_ALIVE)


if __name__ == "__main__":
    main()
