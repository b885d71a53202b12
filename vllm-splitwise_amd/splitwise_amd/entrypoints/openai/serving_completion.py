"""``POST /v1/completions`` (vllm/entrypoints/openai/serving_completion.py:24-382): one or several
prompts (text or token ids), ``n`` choices each, ``echo``, ``logprobs``, plain and server-sent-event
streaming with a final usage count.  Choice ``p * n + i`` is the i-th sample of prompt p.  Every
choice also carries ``token_ids`` (not in the OpenAI schema).
"""
import json
import time
import uuid
from typing import AsyncIterator, List, Union

from fastapi import Request
from fastapi.responses import JSONResponse, StreamingResponse

from .serving_engine import OpenAIServing, RequestError, error_response, merge_streams, usage


def parse_prompts(raw) -> List[Union[str, List[int]]]:
    """The four prompt forms of the completions API (serving_completion.py:206-232)."""
    if isinstance(raw, str):
        return [raw]
    if isinstance(raw, list) and raw:
        if all(isinstance(t, int) for t in raw):
            return [raw]
        if all(isinstance(p, str) for p in raw) or all(isinstance(p, list) for p in raw):
            return list(raw)
    raise RequestError("prompt must be a string, a list of strings, a list of token ids, or a list of such lists")


class OpenAIServingCompletion(OpenAIServing):
    async def create_completion(self, body: dict, raw_request: Request):
        try:
            self.check_model(body)
            prompts = parse_prompts(body.get("prompt"))
            n = self.num_choices(body)
            want = body.get("logprobs")
            echo = bool(body.get("echo", False))
            stream = bool(body.get("stream", False))
            if echo and body.get("max_tokens") == 0:
                raise RequestError("echo without generation (max_tokens = 0) is not supported by this server")
            request_id = f"cmpl-{uuid.uuid4().hex}"
            jobs = []                               # (choice index, prompt text or None, prompt ids, stream)
            for p, prompt in enumerate(prompts):
                for i in range(n):
                    params = self.sampling_params(body, 1.0, 16, want)
                    if params.seed is not None:
                        params.seed += i            # the n samples of a seeded request differ, reproducibly
                    ids = self.tokenize(prompt, params)
                    jobs.append((p * n + i, prompt if isinstance(prompt, str) else None, ids,
                                 self.engine.generate(None, params, f"{request_id}-{p * n + i}", prompt_token_ids=ids)))
        except RequestError as e:
            return error_response(str(e), e.status, e.kind)
        created = int(time.time())
        ids_of = [f"{request_id}-{j[0]}" for j in jobs]

        def prompt_text(job) -> str:
            if job[1] is not None:
                return job[1]
            return self.tokenizer.decode(job[2]) if self.tokenizer is not None else ""

        def choice(index: int, text: str, token_ids, logprobs, finish_reason) -> dict:
            # "token_ids" is an extension: the generated ids (of this chunk, when streaming), for clients of
            # a server that runs without a tokenizer
            return {"index": index, "text": text, "token_ids": token_ids, "logprobs": logprobs,
                    "finish_reason": finish_reason}

        if stream:
            return StreamingResponse(self._stream(jobs, ids_of, request_id, created, echo, want, prompt_text, choice),
                                     media_type="text/event-stream")

        finals = [None] * len(jobs)
        try:
            async for k, out in merge_streams([j[3] for j in jobs]):
                if await raw_request.is_disconnected():
                    await self.engine.abort(ids_of)
                    return error_response("Client disconnected")
                finals[k] = out
        except (ValueError, NotImplementedError) as e:
            await self.engine.abort(ids_of)
            return error_response(str(e))
        choices, n_prompt, n_out = [], 0, 0
        for job, final in zip(jobs, finals):
            o = final.outputs[0]
            text = (prompt_text(job) if echo else "") + o.text
            lp = None
            if want is not None:
                tokens, tops = list(o.token_ids), list(o.logprobs or [])
                if echo:                            # prompt tokens carry no log-probability here
                    tokens, tops = list(job[2]) + tokens, [None] * len(job[2]) + tops
                lp = self.create_logprobs(tokens, tops, want)
            choices.append(choice(job[0], text, list(o.token_ids), lp, o.finish_reason))
            n_prompt += len(final.prompt_token_ids)
            n_out += len(o.token_ids)
        return JSONResponse({"id": request_id, "object": "text_completion", "created": created,
                             "model": self.served_model, "choices": choices, "usage": usage(n_prompt, n_out)})

    async def _stream(self, jobs, ids_of, request_id, created, echo, want, prompt_text, choice) -> AsyncIterator[str]:
        def event(choices, use=None) -> str:
            body = {"id": request_id, "object": "text_completion", "created": created, "model": self.served_model,
                    "choices": choices}
            if use is not None:
                body["usage"] = use
            return f"data: {json.dumps(body, ensure_ascii=False)}\n\n"

        sent_chars = [0] * len(jobs)
        sent_tokens = [0] * len(jobs)
        n_prompt = n_out = 0
        try:
            async for k, out in merge_streams([j[3] for j in jobs]):
                o = out.outputs[0]
                delta = o.text[sent_chars[k]:]
                offset = sent_chars[k]
                if echo and sent_tokens[k] == 0 and sent_chars[k] == 0:
                    head = prompt_text(jobs[k])
                    delta, offset = head + delta, 0
                lp = None
                if want is not None:
                    lp = self.create_logprobs(list(o.token_ids[sent_tokens[k]:]), list((o.logprobs or [])[sent_tokens[k]:]),
                                              want, initial_text_offset=offset)
                new_ids = list(o.token_ids[sent_tokens[k]:])
                sent_chars[k], sent_tokens[k] = len(o.text), len(o.token_ids)
                yield event([choice(jobs[k][0], delta, new_ids, lp, o.finish_reason if out.finished else None)])
                if out.finished:
                    n_prompt += len(out.prompt_token_ids)
                    n_out += len(o.token_ids)
            # the count of what was used closes the stream (serving_completion.py:136-151)
            yield event([], usage(n_prompt, n_out))
        except (ValueError, NotImplementedError) as e:
            await self.engine.abort(ids_of)
            yield f"data: {json.dumps({'object': 'error', 'message': str(e), 'type': 'BadRequestError', 'code': 400})}\n\n"
        yield "data: [DONE]\n\n"
