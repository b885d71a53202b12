"""What the completion and chat front-ends share (vllm/entrypoints/openai/serving_engine.py:21-191):
model-name check, request → SamplingParams, prompt tokenisation with the context-length check, the
OpenAI ``logprobs`` object, error bodies, and fan-in of several generation streams.

Requests and responses are plain JSON objects with the OpenAI field names; a field this build does
not implement (penalties, logit_bias, beam search, guided decoding, suffix) is refused with a 400
when it is set to something other than its neutral value, never ignored silently.
"""
import asyncio
from http import HTTPStatus
from typing import AsyncIterator, Dict, List, Optional, Tuple, Union

from fastapi.responses import JSONResponse

from ...engine.async_llm_engine import AsyncLLMEngine
from ...outputs import RequestOutput
from ...sampling_params import SamplingParams
from ...sequence import Logprob

# field -> neutral value: anything else is a feature outside this build
_UNSUPPORTED = {"presence_penalty": 0.0, "frequency_penalty": 0.0, "repetition_penalty": 1.0, "min_p": 0.0,
                "length_penalty": 1.0, "logit_bias": None, "use_beam_search": False, "early_stopping": False,
                "guided_json": None, "guided_regex": None, "guided_choice": None, "suffix": None}


class RequestError(ValueError):
    def __init__(self, message: str, status: HTTPStatus = HTTPStatus.BAD_REQUEST,
                 kind: str = "BadRequestError") -> None:
        super().__init__(message)
        self.status, self.kind = status, kind


def error_response(message: str, status: HTTPStatus = HTTPStatus.BAD_REQUEST,
                   kind: str = "BadRequestError") -> JSONResponse:
    return JSONResponse({"object": "error", "message": message, "type": kind, "param": None, "code": status.value},
                        status_code=status.value)


def usage(prompt_tokens: int, completion_tokens: int) -> Dict[str, int]:
    return {"prompt_tokens": prompt_tokens, "completion_tokens": completion_tokens,
            "total_tokens": prompt_tokens + completion_tokens}


class OpenAIServing:
    def __init__(self, engine: AsyncLLMEngine, served_model: str) -> None:
        self.engine = engine
        self.served_model = served_model
        self.tokenizer = engine.engine.tokenizer
        self.max_model_len = engine.engine.model_config.max_model_len

    # ------------------------------------------------------------------ checks
    def check_model(self, body: dict) -> None:
        model = body.get("model")
        if model is not None and model != self.served_model:
            raise RequestError(f"The model `{model}` does not exist.", HTTPStatus.NOT_FOUND, "NotFoundError")

    def show_available_models(self) -> dict:
        import time
        return {"object": "list", "data": [{"id": self.served_model, "object": "model", "created": int(time.time()),
                                            "owned_by": "splitwise_amd", "root": self.served_model, "parent": None,
                                            "max_model_len": self.max_model_len}]}

    @staticmethod
    def num_choices(body: dict) -> int:
        n = body.get("n")
        if n is None:
            return 1
        if not isinstance(n, int) or isinstance(n, bool) or n < 1:
            raise RequestError(f"n must be an integer of at least 1, got {n!r}.")
        return n

    @staticmethod
    def sampling_params(body: dict, default_temperature: float, default_max_tokens: Optional[int],
                        logprobs: Optional[int]) -> SamplingParams:
        """Request fields → SamplingParams for ONE sequence (protocol.py:93-149,187-242); ``n`` is served
        as that many single-sequence requests by the callers."""
        for name, neutral in _UNSUPPORTED.items():
            value = body.get(name)
            if value is not None and value != neutral and value != {}:
                raise RequestError(f"{name} is not supported by this server")
        best_of = body.get("best_of")
        if best_of is not None and best_of != OpenAIServing.num_choices(body):
            raise RequestError("best_of different from n is not supported by this server")
        stop = body.get("stop")
        try:
            return SamplingParams(
                n=1, temperature=default_temperature if body.get("temperature") is None else body["temperature"],
                top_p=body.get("top_p") if body.get("top_p") is not None else 1.0,
                top_k=body.get("top_k") if body.get("top_k") is not None else -1,
                stop=stop if stop else None, stop_token_ids=body.get("stop_token_ids") or None,
                ignore_eos=bool(body.get("ignore_eos", False)),
                max_tokens=body.get("max_tokens", default_max_tokens), logprobs=logprobs,
                skip_special_tokens=body.get("skip_special_tokens", True),
                spaces_between_special_tokens=body.get("spaces_between_special_tokens", True),
                include_stop_str_in_output=bool(body.get("include_stop_str_in_output", False)),
                seed=body.get("seed"))
        except (TypeError, ValueError, NotImplementedError) as e:
            raise RequestError(str(e)) from e

    def tokenize(self, prompt: Union[str, List[int]], params: SamplingParams) -> List[int]:
        """Token ids of one prompt; settles ``max_tokens`` against the context length
        (serving_engine.py:160-191)."""
        if isinstance(prompt, str):
            if self.tokenizer is None:
                raise RequestError("this server runs without a tokenizer: send token ids as the prompt")
            ids = list(self.tokenizer(prompt).input_ids)
        else:
            ids = list(prompt)
            if not ids or not all(isinstance(t, int) and t >= 0 for t in ids):
                raise RequestError("a prompt given as token ids must be a non-empty list of non-negative integers")
        if params.max_tokens is None:
            params.max_tokens = self.max_model_len - len(ids)
        if len(ids) + (params.max_tokens or 0) > self.max_model_len or params.max_tokens < 1:
            raise RequestError(
                f"This model's maximum context length is {self.max_model_len} tokens. However, you requested "
                f"{len(ids) + max(params.max_tokens, 0)} tokens ({len(ids)} in the messages, "
                f"{max(params.max_tokens, 0)} in the completion). Please reduce the length of the messages or completion.")
        return ids

    # ---------------------------------------------------------------- logprobs
    def token_text(self, token_id: int, entry: Optional[Logprob] = None) -> str:
        if entry is not None and entry.decoded_token is not None:
            return entry.decoded_token
        if self.tokenizer is None:
            return str(token_id)
        return self.tokenizer.convert_ids_to_tokens(token_id)

    def create_logprobs(self, token_ids: List[int], top_logprobs: List[Optional[Dict[int, Logprob]]],
                        num_output_top_logprobs: Optional[int] = None, initial_text_offset: int = 0) -> dict:
        """The completions API's ``logprobs`` object (serving_engine.py:59-93)."""
        out = {"text_offset": [], "token_logprobs": [], "tokens": [],
               "top_logprobs": [] if num_output_top_logprobs else None}
        offset = initial_text_offset
        for i, token_id in enumerate(token_ids):
            step = top_logprobs[i] if i < len(top_logprobs) else None
            entry = step.get(token_id) if step else None
            text = self.token_text(token_id, entry)
            out["tokens"].append(text)
            out["token_logprobs"].append(None if entry is None else entry.logprob)
            out["text_offset"].append(offset)
            offset += len(text)
            if num_output_top_logprobs:
                out["top_logprobs"].append(None if not step else
                                           {self.token_text(t, e): e.logprob for t, e in step.items()})
        return out


async def merge_streams(streams: List[AsyncIterator[RequestOutput]]) -> AsyncIterator[Tuple[int, RequestOutput]]:
    """(index, item) pairs of several async generators in arrival order; the first exception of any of
    them is raised after the others have been cancelled."""
    queue: asyncio.Queue = asyncio.Queue()
    done = object()

    async def pump(i: int, stream) -> None:
        try:
            async for item in stream:
                await queue.put((i, item))
            await queue.put((i, done))
        except BaseException as e:                  # noqa: BLE001 - handed to the consumer
            await queue.put((i, e))

    tasks = [asyncio.ensure_future(pump(i, s)) for i, s in enumerate(streams)]
    live = len(tasks)
    try:
        while live:
            i, item = await queue.get()
            if item is done:
                live -= 1
            elif isinstance(item, BaseException):
                raise item
            else:
                yield i, item
    finally:
        for t in tasks:
            t.cancel()
