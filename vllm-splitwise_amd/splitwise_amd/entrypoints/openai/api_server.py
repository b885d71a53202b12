"""OpenAI-style HTTP front (vllm/entrypoints/openai/api_server.py:36-265): ``/health``, ``/metrics``
(Prometheus), ``/v1/models``, ``/v1/completions``, ``/v1/chat/completions``; optional bearer key,
CORS settings, chat template and response role from the command line.
"""
import argparse
import json
from typing import List, Optional

from fastapi import FastAPI, Request
from fastapi.exceptions import RequestValidationError
from fastapi.middleware.cors import CORSMiddleware
from fastapi.responses import JSONResponse, Response

from ...engine.arg_utils import EngineArgs
from ...engine.async_llm_engine import AsyncLLMEngine
from .serving_chat import OpenAIServingChat
from .serving_completion import OpenAIServingCompletion
from .serving_engine import error_response


def build_app(engine: AsyncLLMEngine, served_model: str, response_role: str = "assistant",
              chat_template: Optional[str] = None, api_key: Optional[str] = None, root_path: Optional[str] = None,
              allowed_origins: Optional[List[str]] = None, allow_credentials: bool = False,
              allowed_methods: Optional[List[str]] = None, allowed_headers: Optional[List[str]] = None) -> FastAPI:
    app = FastAPI(root_path=root_path or "")
    app.add_middleware(CORSMiddleware, allow_origins=allowed_origins or ["*"], allow_credentials=allow_credentials,
                       allow_methods=allowed_methods or ["*"], allow_headers=allowed_headers or ["*"])
    completion = OpenAIServingCompletion(engine, served_model)
    chat = OpenAIServingChat(engine, served_model, response_role, chat_template)

    try:                                            # /metrics (openai/api_server.py:150-152)
        from prometheus_client import make_asgi_app
        from ...engine import metrics
        metrics.ensure_registered(served_model)     # the vllm:* series exist before the first step
        app.mount("/metrics", make_asgi_app())
    except ImportError:                             # pragma: no cover - the front works without it
        pass

    if api_key:                                     # openai/api_server.py:215-229
        @app.middleware("http")
        async def authentication(request: Request, call_next):
            prefix = (root_path or "") + "/v1"
            if request.url.path.startswith(prefix) and request.headers.get("Authorization") != "Bearer " + api_key:
                return JSONResponse({"error": "Unauthorized"}, status_code=401)
            return await call_next(request)

    @app.exception_handler(RequestValidationError)
    async def validation_exception_handler(_, exc):
        return error_response(str(exc))

    async def json_body(request: Request):
        try:
            body = await request.json()
        except (json.JSONDecodeError, UnicodeDecodeError) as e:
            return None, error_response(f"request body is not valid JSON: {e}")
        if not isinstance(body, dict):
            return None, error_response("request body must be a JSON object")
        return body, None

    @app.get("/health")
    async def health() -> Response:
        await engine.check_health()
        return Response(status_code=200)

    @app.get("/v1/models")
    async def show_available_models() -> JSONResponse:
        return JSONResponse(completion.show_available_models())

    @app.post("/v1/completions")
    async def create_completion(request: Request):
        body, err = await json_body(request)
        return err if err is not None else await completion.create_completion(body, request)

    @app.post("/v1/chat/completions")
    async def create_chat_completion(request: Request):
        body, err = await json_body(request)
        return err if err is not None else await chat.create_chat_completion(body, request)

    return app


def make_arg_parser() -> argparse.ArgumentParser:
    """The reference's front-end flags (openai/api_server.py:57-139), then the engine's."""
    parser = argparse.ArgumentParser(description="OpenAI-compatible RESTful API server.")
    parser.add_argument("--host", type=str, default=None)
    parser.add_argument("--port", type=int, default=8000)
    parser.add_argument("--allow-credentials", action="store_true")
    parser.add_argument("--allowed-origins", type=json.loads, default=["*"])
    parser.add_argument("--allowed-methods", type=json.loads, default=["*"])
    parser.add_argument("--allowed-headers", type=json.loads, default=["*"])
    parser.add_argument("--api-key", type=str, default=None,
                        help="If provided, the server will require this key to be presented in the header.")
    parser.add_argument("--served-model-name", type=str, default=None)
    parser.add_argument("--chat-template", type=str, default=None,
                        help="File path of a chat template, or the template in single-line form.")
    parser.add_argument("--response-role", type=str, default="assistant")
    parser.add_argument("--ssl-keyfile", type=str, default=None)
    parser.add_argument("--ssl-certfile", type=str, default=None)
    parser.add_argument("--root-path", type=str, default=None)
    return EngineArgs.add_cli_args(parser)


def main() -> None:
    import uvicorn
    args = make_arg_parser().parse_args()
    engine = AsyncLLMEngine.from_engine_args(EngineArgs.from_cli_args(args))
    app = build_app(engine, args.served_model_name or args.model, args.response_role, args.chat_template,
                    args.api_key, args.root_path, args.allowed_origins, args.allow_credentials,
                    args.allowed_methods, args.allowed_headers)
    uvicorn.run(app, host=args.host, port=args.port, log_level="info", timeout_keep_alive=5,
                ssl_keyfile=args.ssl_keyfile, ssl_certfile=args.ssl_certfile)


if __name__ == "__main__":
    main()
