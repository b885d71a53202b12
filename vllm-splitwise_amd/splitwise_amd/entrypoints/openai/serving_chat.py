"""``POST /v1/chat/completions`` (vllm/entrypoints/openai/serving_chat.py:20-332): the conversation
is rendered with the tokenizer's chat template (or one given on the command line), ``n`` choices,
plain and streaming answers (a role chunk first, then content deltas, then the finish reason),
``echo`` of the last message, usage counts.
"""
import codecs
import json
import os
import time
import uuid
from typing import AsyncIterator, Optional

from fastapi import Request
from fastapi.responses import JSONResponse, StreamingResponse

from .serving_engine import OpenAIServing, RequestError, error_response, merge_streams, usage


class OpenAIServingChat(OpenAIServing):
    def __init__(self, engine, served_model: str, response_role: str = "assistant",
                 chat_template: Optional[str] = None) -> None:
        super().__init__(engine, served_model)
        self.response_role = response_role
        self._load_chat_template(chat_template)

    def _load_chat_template(self, chat_template: Optional[str]) -> None:
        """A file name, or the template itself with escaped newlines (serving_chat.py:313-332)."""
        if chat_template is None or self.tokenizer is None:
            return
        if os.path.isfile(chat_template):
            with open(chat_template, "r") as f:
                self.tokenizer.chat_template = f.read()
        else:
            self.tokenizer.chat_template = codecs.decode(chat_template, "unicode_escape")

    def get_chat_request_role(self, body: dict) -> str:
        messages = body.get("messages") or []
        if body.get("add_generation_prompt", True) or not messages:
            return self.response_role
        return messages[-1]["role"]

    async def create_chat_completion(self, body: dict, raw_request: Request):
        try:
            self.check_model(body)
            if self.tokenizer is None or getattr(self.tokenizer, "chat_template", None) is None:
                raise RequestError("chat completions need a tokenizer with a chat template; start the server with "
                                   "--tokenizer (and --chat-template), or send token ids to /v1/completions")
            messages = body.get("messages")
            if not isinstance(messages, list) or not messages or not all(
                    isinstance(m, dict) and "role" in m and "content" in m for m in messages):
                raise RequestError("messages must be a non-empty list of {role, content} objects")
            try:
                prompt = self.tokenizer.apply_chat_template(conversation=messages, tokenize=False,
                                                            add_generation_prompt=body.get("add_generation_prompt", True))
            except Exception as e:                   # noqa: BLE001 - a template error is the caller's
                raise RequestError(f"Error in applying chat template from request: {e}") from e
            n = self.num_choices(body)
            top = body.get("top_logprobs") if body.get("logprobs") else None
            if body.get("logprobs") and top is None:
                top = 0
            request_id = f"chatcmpl-{uuid.uuid4().hex}"
            jobs = []
            for i in range(n):
                params = self.sampling_params(body, 0.7, None, top)
                if params.seed is not None:
                    params.seed += i
                ids = self.tokenize(prompt, params)
                jobs.append((i, self.engine.generate(None, params, f"{request_id}-{i}", prompt_token_ids=ids)))
        except RequestError as e:
            return error_response(str(e), e.status, e.kind)
        created = int(time.time())
        role = self.get_chat_request_role(body)
        ids_of = [f"{request_id}-{i}" for i, _ in jobs]
        echo_text = ""
        if body.get("echo", False) and messages[-1].get("role") == role:
            echo_text = messages[-1].get("content") or ""

        if body.get("stream", False):
            return StreamingResponse(self._stream(jobs, ids_of, request_id, created, role, echo_text, top),
                                     media_type="text/event-stream")
        finals = [None] * len(jobs)
        try:
            async for k, out in merge_streams([g for _, g in jobs]):
                if await raw_request.is_disconnected():
                    await self.engine.abort(ids_of)
                    return error_response("Client disconnected")
                finals[k] = out
        except (ValueError, NotImplementedError) as e:
            await self.engine.abort(ids_of)
            return error_response(str(e))
        choices, n_out = [], 0
        for (i, _), final in zip(jobs, finals):
            o = final.outputs[0]
            lp = self.create_logprobs(list(o.token_ids), list(o.logprobs or []), top) if top is not None else None
            choices.append({"index": i, "message": {"role": role, "content": echo_text + o.text}, "logprobs": lp,
                            "finish_reason": o.finish_reason})
            n_out += len(o.token_ids)
        return JSONResponse({"id": request_id, "object": "chat.completion", "created": created,
                             "model": self.served_model, "choices": choices,
                             "usage": usage(len(finals[0].prompt_token_ids), n_out)})

    async def _stream(self, jobs, ids_of, request_id, created, role, echo_text, top) -> AsyncIterator[str]:
        def event(choices, use=None) -> str:
            body = {"id": request_id, "object": "chat.completion.chunk", "created": created,
                    "model": self.served_model, "choices": choices}
            if use is not None:
                body["usage"] = use
            return f"data: {json.dumps(body, ensure_ascii=False)}\n\n"

        # the role of every choice first, then the echoed message if asked for (serving_chat.py:97-137)
        for i, _ in jobs:
            yield event([{"index": i, "delta": {"role": role}, "logprobs": None, "finish_reason": None}])
            if echo_text:
                yield event([{"index": i, "delta": {"content": echo_text}, "logprobs": None, "finish_reason": None}])
        sent_chars = [0] * len(jobs)
        sent_tokens = [0] * len(jobs)
        n_prompt = n_out = 0
        try:
            async for k, out in merge_streams([g for _, g in jobs]):
                o = out.outputs[0]
                delta = o.text[sent_chars[k]:]
                lp = None
                if top is not None:
                    lp = self.create_logprobs(list(o.token_ids[sent_tokens[k]:]), list((o.logprobs or [])[sent_tokens[k]:]),
                                              top, initial_text_offset=sent_chars[k])
                sent_chars[k], sent_tokens[k] = len(o.text), len(o.token_ids)
                if delta or not out.finished:
                    yield event([{"index": jobs[k][0], "delta": {"content": delta}, "logprobs": lp,
                                  "finish_reason": None}])
                if out.finished:
                    n_prompt = len(out.prompt_token_ids)
                    n_out += len(o.token_ids)
                    yield event([{"index": jobs[k][0], "delta": {}, "logprobs": None, "finish_reason": o.finish_reason}],
                                usage(n_prompt, len(o.token_ids)))
        except (ValueError, NotImplementedError) as e:
            await self.engine.abort(ids_of)
            yield f"data: {json.dumps({'object': 'error', 'message': str(e), 'type': 'BadRequestError', 'code': 400})}\n\n"
        yield "data: [DONE]\n\n"
