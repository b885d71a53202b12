"""Asynchronous front of the engine (vllm/engine/async_llm_engine.py:24-690).

One background task drives ``LLMEngine.step`` whenever requests are in flight; callers iterate
``generate(...)`` and receive every ``RequestOutput`` of their request.  The step itself runs on
a worker thread so the event loop keeps accepting requests while the GPUs are busy (the
reference does the same for its driver worker, async_llm_engine.py:197-218).
"""
import asyncio
import time
from typing import AsyncIterator, Dict, Iterable, List, Optional, Set, Tuple, Union

from ..outputs import RequestOutput
from ..sampling_params import SamplingParams
from .arg_utils import EngineArgs
from .llm_engine import LLMEngine


class AsyncEngineDeadError(RuntimeError):
    pass


class AsyncStream:
    """Outputs of one request, as an async iterator (async_llm_engine.py:47-76)."""

    def __init__(self, request_id: str) -> None:
        self.request_id = request_id
        self._queue: asyncio.Queue = asyncio.Queue()
        self._finished = False

    def put(self, item: Union[RequestOutput, Exception]) -> None:
        if not self._finished:
            self._queue.put_nowait(item)

    def finish(self) -> None:
        self._queue.put_nowait(StopAsyncIteration())
        self._finished = True

    @property
    def finished(self) -> bool:
        return self._finished

    def __aiter__(self):
        return self

    async def __anext__(self) -> RequestOutput:
        item = await self._queue.get()
        if isinstance(item, Exception):
            raise item
        return item


class RequestTracker:
    """Book-keeping between callers and the engine loop (async_llm_engine.py:79-191)."""

    def __init__(self) -> None:
        self._streams: Dict[str, AsyncStream] = {}
        self._new: asyncio.Queue = asyncio.Queue()
        self._aborted: asyncio.Queue = asyncio.Queue()
        self.new_requests_event = asyncio.Event()

    def __contains__(self, request_id: str) -> bool:
        return request_id in self._streams

    def __len__(self) -> int:
        return len(self._streams)

    def propagate_exception(self, exc: Exception, request_id: Optional[str] = None) -> None:
        targets = [request_id] if request_id is not None else list(self._streams)
        for rid in targets:
            if rid in self._streams:
                self._streams[rid].put(exc)
                self.abort_request(rid)

    def process_request_output(self, output: RequestOutput) -> None:
        stream = self._streams.get(output.request_id)
        if stream is None:
            return
        stream.put(output)
        if output.finished:
            self.abort_request(output.request_id)

    def add_request(self, request_id: str, **engine_add_request_kwargs) -> AsyncStream:
        if request_id in self._streams:
            raise KeyError(f"Request {request_id} already exists.")
        stream = AsyncStream(request_id)
        self._new.put_nowait((stream, {"request_id": request_id, **engine_add_request_kwargs}))
        self.new_requests_event.set()
        return stream

    def abort_request(self, request_id: str) -> None:
        self._aborted.put_nowait(request_id)
        stream = self._streams.get(request_id)
        if stream is not None and not stream.finished:
            stream.finish()

    def get_new_and_finished_requests(self) -> Tuple[List[dict], Set[str]]:
        finished: Set[str] = set()
        while not self._aborted.empty():
            rid = self._aborted.get_nowait()
            finished.add(rid)
            self._streams.pop(rid, None)
        new: List[dict] = []
        while not self._new.empty():
            stream, kwargs = self._new.get_nowait()
            if stream.request_id in finished:
                stream.finish()
                continue
            self._streams[stream.request_id] = stream
            new.append(kwargs)
        self.new_requests_event.clear()
        return new, finished

    async def wait_for_new_requests(self) -> None:
        await self.new_requests_event.wait()


class AsyncLLMEngine:
    def __init__(self, *engine_args, start_engine_loop: bool = True, engine: Optional[LLMEngine] = None,
                 **engine_kwargs) -> None:
        self.engine = engine if engine is not None else LLMEngine(*engine_args, **engine_kwargs)
        self.start_engine_loop = start_engine_loop
        self.background_loop: Optional[asyncio.Task] = None
        self._request_tracker: Optional[RequestTracker] = None
        self._errored_with: Optional[BaseException] = None

    @classmethod
    def from_engine_args(cls, engine_args: EngineArgs, start_engine_loop: bool = True) -> "AsyncLLMEngine":
        return cls(engine=LLMEngine.from_engine_args(engine_args), start_engine_loop=start_engine_loop)

    # ------------------------------------------------------------------ state
    @property
    def is_running(self) -> bool:
        return self.background_loop is not None and not self.background_loop.done()

    @property
    def errored(self) -> bool:
        return self._errored_with is not None

    def start_background_loop(self) -> None:
        if self.errored:
            raise AsyncEngineDeadError("Background loop has errored already.") from self._errored_with
        if self.is_running:
            raise RuntimeError("Background loop is already running.")
        self._request_tracker = RequestTracker()
        self.background_loop = asyncio.get_event_loop().create_task(self.run_engine_loop())
        self.background_loop.add_done_callback(self._on_loop_done)

    def _on_loop_done(self, task: asyncio.Task) -> None:
        if task.cancelled():
            return
        exc = task.exception()
        if exc is not None:
            self._errored_with = exc
            if self._request_tracker is not None:
                dead = AsyncEngineDeadError(f"engine loop failed: {exc!r}")
                dead.__cause__ = exc                 # waiting requests see where the loop died, not only that it did
                self._request_tracker.propagate_exception(dead)

    # ------------------------------------------------------------------- loop
    async def engine_step(self) -> bool:
        """Admits new requests, drops aborted ones, runs one step.  -> whether work remains."""
        new, finished = self._request_tracker.get_new_and_finished_requests()
        for kwargs in new:
            try:
                self.engine.add_request(**kwargs)
            except (ValueError, NotImplementedError) as e:
                self._request_tracker.propagate_exception(e, kwargs["request_id"])
        if finished:
            self.engine.abort_request(finished)
        if not self.engine.has_unfinished_requests():
            return False
        outputs = await asyncio.get_event_loop().run_in_executor(None, self.engine.step)
        for out in outputs:
            self._request_tracker.process_request_output(out)
        return self.engine.has_unfinished_requests() or len(outputs) > 0

    async def run_engine_loop(self) -> None:
        has_work = False
        while True:
            if not has_work:
                await self._request_tracker.wait_for_new_requests()
            has_work = await self.engine_step()
            await asyncio.sleep(0)

    # --------------------------------------------------------------- requests
    async def add_request(self, request_id: str, prompt: Optional[str], sampling_params: SamplingParams,
                          prompt_token_ids: Optional[List[int]] = None,
                          arrival_time: Optional[float] = None) -> AsyncStream:
        if not self.is_running:
            if self.start_engine_loop:
                self.start_background_loop()
            else:
                raise AsyncEngineDeadError("Background loop is not running. If it was running, inspect the output "
                                           "to find the stacktrace of the error that caused the background loop "
                                           "to stop (AsyncEngineDeadError).")
        if arrival_time is None:
            arrival_time = time.monotonic()
        return self._request_tracker.add_request(request_id, prompt=prompt, sampling_params=sampling_params,
                                                 prompt_token_ids=prompt_token_ids, arrival_time=arrival_time)

    async def generate(self, prompt: Optional[str], sampling_params: SamplingParams, request_id: str,
                       prompt_token_ids: Optional[List[int]] = None) -> AsyncIterator[RequestOutput]:
        """Yields the request's outputs as the engine produces them (async_llm_engine.py:538-628)."""
        stream = await self.add_request(request_id, prompt, sampling_params, prompt_token_ids=prompt_token_ids,
                                        arrival_time=time.monotonic())
        try:
            async for out in stream:
                yield out
        except (Exception, asyncio.CancelledError):
            self._abort(request_id)
            raise

    async def abort(self, request_id: Union[str, Iterable[str]]) -> None:
        if not self.is_running:
            raise AsyncEngineDeadError("Background loop is not running.")
        for rid in ([request_id] if isinstance(request_id, str) else request_id):
            self._abort(rid)

    def _abort(self, request_id: str) -> None:
        if self._request_tracker is not None:
            self._request_tracker.abort_request(request_id)

    async def get_model_config(self):
        return self.engine.model_config

    async def check_health(self) -> None:
        if self.errored:
            raise AsyncEngineDeadError("Background loop is stopped.") from self._errored_with
        self.engine.check_health()
