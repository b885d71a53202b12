"""The synchronous engine: request intake, one scheduler + executor step, output processing.

API and step structure of vllm/engine/llm_engine.py:150-633 (from_engine_args, add_request,
abort_request, step, has_unfinished_requests); stop logic of _check_stop (:755-784).  Beam
search and n > 1 forking are outside this build (one sequence per request).
"""
import gc
import os
import time
from collections import deque
from typing import Deque, Dict, Iterable, List, Optional, Union

from ..config import CacheConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig
from ..core.scheduler import Scheduler, SchedulerGroup, SchedulerOutputs
from ..outputs import RequestOutput
from ..sampling_params import SamplingParams
from ..sequence import (SamplerOutput, Sequence, SequenceGroup, SequenceGroupMetadata, SequenceStatus)
from ..transformers_utils.tokenizer import detokenize_incrementally, get_tokenizer
from ..utils import coalesce_block_pairs
from .arg_utils import EngineArgs
from .metrics import StatLogger, Stats


class LLMEngine:
    def __init__(self, model_config: ModelConfig, cache_config: CacheConfig, parallel_config: ParallelConfig,
                 scheduler_config: SchedulerConfig, device_config: DeviceConfig, executor_class=None,
                 log_stats: bool = False, tokenizer=None, num_gpu_blocks_override: Optional[int] = None,
                 executor=None) -> None:
        self._freeze_gc = os.environ.get("SPLITWISE_FREEZE_GC", "1") != "0"
        if self._freeze_gc:
            gc.unfreeze()       # what an earlier engine of this process froze may be garbage by now (see the end)
        self.model_config = model_config
        self.cache_config = cache_config
        self.parallel_config = parallel_config
        self.scheduler_config = scheduler_config
        self.device_config = device_config
        self.log_stats = log_stats
        self.tokenizer = tokenizer
        self.seq_counter = 0
        if executor is not None:
            self.model_executor = executor
        else:
            if executor_class is None:
                if getattr(parallel_config, "one_gpu_overlap", False):
                    from ..executor.overlap_executor import OverlapExecutor
                    executor_class = OverlapExecutor
                elif parallel_config.tensor_parallel_size > 1:
                    from ..executor.pool_executor import PoolExecutor
                    executor_class = PoolExecutor
                elif parallel_config.world_size > 1:
                    from ..executor.splitwise_executor import SplitwiseExecutor
                    executor_class = SplitwiseExecutor
                else:
                    from ..executor.gpu_executor import GPUExecutor
                    executor_class = GPUExecutor
            self.model_executor = executor_class(model_config, cache_config, parallel_config, scheduler_config,
                                                 device_config, num_gpu_blocks_override=num_gpu_blocks_override)
        # one scheduler -- one block space, one set of hand-off slots -- per token pool ("lane"); the
        # reference's world has one (vllm/config.py:434-438)
        lanes = max(getattr(parallel_config, "num_token_pools", 1), 1)
        # (prompt steps beside decode steps on ONE GPU: the split loop over a single cache, so what a decode step
        # wrote is there for a later prompt step to read, as in the co-located engine)
        self._decode_kv_shareable = (not parallel_config.sep_prompt_token
                                     or bool(getattr(parallel_config, "one_gpu_overlap", False)))
        self.schedulers = [Scheduler(scheduler_config, cache_config, None,
                                     track_prompt_blocks=parallel_config.sep_prompt_token,
                                     one_cache=self._decode_kv_shareable) for _ in range(lanes)]
        self.scheduler = self.schedulers[0] if lanes == 1 else SchedulerGroup(self.schedulers)
        # a prompt pool that keeps only the prompts in flight hands out its own block numbers, step by step
        self.staged_prompt_cache = bool(getattr(parallel_config, "staged_prompt_cache", False))
        self._staging_blocks = getattr(self.model_executor, "prompt_cache_blocks", None)
        if self.staged_prompt_cache and not self._staging_blocks:
            raise ValueError("staged_prompt_cache needs an executor that sizes the prompt pool's cache by itself")
        self.stat_logger = None
        if log_stats:
            self.stat_logger = StatLogger(local_interval=5.0, labels=dict(model_name=model_config.model))
            self.stat_logger.info("cache_config", cache_config)
        # split pools: run a prompt step and a decode step concurrently when the executor can
        self.overlap_pools = parallel_config.sep_prompt_token and hasattr(self.model_executor, "execute_split")
        self._slot_epoch: Dict[tuple, int] = {}        # (lane, hand-off slot) -> hand-offs so far
        # pipelined steps (co-located engine only): step n+1 is scheduled and launched while step n
        # still runs; a step's tokens are appended, and its outputs returned, one call later
        self.pipeline_steps = (getattr(scheduler_config, "pipeline_steps", False) and not self.overlap_pools
                               and getattr(self.model_executor, "supports_deferred_samples", False))
        self._inflight = None           # (pending samples, scheduler outputs, [(group, sequence)])
        # split pools, pipelined: the prompt pool and the token pool each have a step in flight
        self.pipeline_split = (getattr(scheduler_config, "pipeline_steps", False) and self.overlap_pools
                               and getattr(self.model_executor, "supports_pipelined_split", False))
        # prompt steps launched and not resolved yet, oldest first: (pending samples, scheduler outputs, request ids,
        # lane).  Two deep: the engine looks at the prompt pool once per decode step, so with one step in flight the
        # pool would sit idle from the end of a prompt step to the next look (a prompt GPU that feeds two token GPUs has
        # no such time to give away); the second step waits on the device, in stream order.  A staged prompt cache has
        # room for exactly two steps.
        self._prompt_queue: Deque[tuple] = deque()
        self._prompt_queue_depth = 2
        self._decode_inflight: List[Optional[tuple]] = [None] * lanes   # per lane: (scheduler outputs, [(group, sequence)])
        self._free_later: List[List[Sequence]] = [[] for _ in range(lanes)]
        # requests with best_of > 1 that are alive: while there are any, a pipelined engine runs plain steps (a
        # fork needs the sampled tokens on the host before the next step can be laid out)
        self._multi_sample_requests: set = set()
        self._next_prompt_lane = 0
        if (lanes > 1 or self.staged_prompt_cache) and not self.pipeline_split:
            raise NotImplementedError("several token pools / a staged prompt cache need pipeline_steps and an "
                                      "executor with launch_prompt / decode_send / decode_collect")
        if self.pipeline_split:
            for sched in self.schedulers:
                sched.in_flight = self._seq_in_flight
        # Everything built so far (modules, parameters, captured graphs, block tables) lives as long as the engine
        # does.  Python's cyclic collector would walk all of it on every full collection -- tens of milliseconds
        # in the middle of a serving loop, 4 % of the closed-loop benchmark's throughput -- so those objects are
        # moved out of its reach; what the steps allocate from now on is collected as usual.
        if self._freeze_gc:
            gc.collect()
            gc.freeze()

    def _seq_in_flight(self, seq: Sequence) -> bool:
        """Pipelined split pools: is the sequence part of a prompt step or a decode step that has been
        launched and not resolved yet?"""
        for entry in self._prompt_queue:
            for group in entry[1].scheduled_seq_groups:
                if any(s.seq_id == seq.seq_id for s in group.get_seqs()):
                    return True
        for inflight in self._decode_inflight:
            if inflight is not None and any(s.seq_id == seq.seq_id for _, s in inflight[1]):
                return True
        return False

    def inflight_request_ids(self) -> List[str]:
        """Request ids of the sequences in steps that have been launched and not collected yet (pipelined
        steps; empty otherwise).  Each of them gets one token from that step."""
        ids: List[str] = []
        if self._inflight is not None:
            ids += [group.request_id for group, _ in self._inflight[2]]
        for inflight in self._decode_inflight:
            if inflight is not None:
                ids += [group.request_id for group, _ in inflight[1]]
        for entry in self._prompt_queue:
            ids += list(entry[2])
        return ids

    @classmethod
    def from_engine_args(cls, engine_args: EngineArgs, **kwargs) -> "LLMEngine":
        configs = engine_args.create_engine_configs()
        if kwargs.get("tokenizer") is None and engine_args.tokenizer:
            kwargs["tokenizer"] = get_tokenizer(engine_args.tokenizer)
        return cls(*configs, num_gpu_blocks_override=engine_args.num_gpu_blocks_override, **kwargs)

    # ----------------------------------------------------------------- requests
    def add_request(self, request_id: str, prompt: Optional[str], sampling_params: SamplingParams,
                    prompt_token_ids: Optional[List[int]] = None, arrival_time: Optional[float] = None) -> None:
        if arrival_time is None:
            arrival_time = time.monotonic()
        if prompt_token_ids is None:
            if self.tokenizer is None:
                raise ValueError("prompt_token_ids must be given: no tokenizer is attached to this engine")
            prompt_token_ids = self.tokenizer.encode(prompt)
        if sampling_params.best_of != 1:
            # several sequences forked from one prompt (llm_engine.py:424-534).  Their blocks are shared copy-on-write
            # inside ONE cache; split pools would have to hand every fork's slot over, which the reference's
            # scheduler does not do either (its slot map only knows sequences that were allocated as prompts)
            if self.parallel_config.sep_prompt_token:
                raise NotImplementedError("n > 1 / best_of > 1 with split prompt / token pools; the HTTP front "
                                          "serves n as n requests")
            self._multi_sample_requests.add(request_id)
        eos = getattr(self.tokenizer, "eos_token_id", None)
        seq = Sequence(self.seq_counter, prompt, list(prompt_token_ids), self.cache_config.block_size, eos)
        self.seq_counter += 1
        self.scheduler.add_seq_group(SequenceGroup(request_id, [seq], sampling_params, arrival_time))

    def abort_request(self, request_id: Union[str, Iterable[str]]) -> None:
        self.scheduler.abort_seq_group(request_id)
        self._multi_sample_requests.difference_update([request_id] if isinstance(request_id, str) else request_id)

    def get_num_unfinished_requests(self) -> int:
        return self.scheduler.get_num_unfinished_seq_groups()

    def has_unfinished_requests(self) -> bool:
        return (self.scheduler.has_unfinished_seqs() or self._inflight is not None
                or bool(self._prompt_queue) or any(d is not None for d in self._decode_inflight)
                or any(self._free_later) or bool(self.scheduler.held_after_abort))

    # --------------------------------------------------------------------- step
    def _decode_sequence(self, seq: Sequence, params: SamplingParams) -> None:
        """Appends the text of the newest token to ``seq.output_text`` (llm_engine.py:859-877)."""
        if self.tokenizer is None:
            return
        new_tokens, new_text, *offsets = detokenize_incrementally(
            self.tokenizer, seq.get_token_ids(), seq.tokens, seq.prefix_offset, seq.read_offset,
            skip_special_tokens=params.skip_special_tokens,
            spaces_between_special_tokens=params.spaces_between_special_tokens)
        if params.logprobs is not None and seq.output_logprobs and seq.output_logprobs[-1]:
            # the text each candidate would have added in the sampled token's place (llm_engine.py:835-857)
            ids = seq.get_token_ids()
            for token_id, entry in seq.output_logprobs[-1].items():
                if token_id == ids[-1]:
                    entry.decoded_token = new_text
                elif entry.decoded_token is None:
                    _, entry.decoded_token, _, _ = detokenize_incrementally(
                        self.tokenizer, ids[:-1] + [token_id], seq.tokens, seq.prefix_offset, seq.read_offset,
                        skip_special_tokens=params.skip_special_tokens,
                        spaces_between_special_tokens=params.spaces_between_special_tokens)
        seq.tokens = new_tokens if seq.tokens is None else seq.tokens + new_tokens
        seq.prefix_offset, seq.read_offset = offsets
        seq.output_text += new_text

    @staticmethod
    def _stop_string_hit(seq: Sequence, params: SamplingParams) -> bool:
        """A stop string ends the output and is cut from it unless asked to stay (llm_engine.py:881-888)."""
        for stop_str in params.stop:
            if stop_str and seq.output_text.endswith(stop_str):
                if not params.include_stop_str_in_output:
                    seq.output_text = seq.output_text[:-len(stop_str)]
                return True
        return False

    def _check_stop(self, seq: Sequence, params: SamplingParams) -> None:
        if self._stop_string_hit(seq, params):
            seq.status = SequenceStatus.FINISHED_STOPPED
            return
        last = seq.get_last_token_id()
        if last in params.stop_token_ids:
            seq.status = SequenceStatus.FINISHED_STOPPED
            return
        if seq.get_len() > self.scheduler_config.max_model_len:
            seq.status = SequenceStatus.FINISHED_LENGTH_CAPPED
            return
        if params.max_tokens is not None and seq.get_output_len() == params.max_tokens:
            seq.status = SequenceStatus.FINISHED_LENGTH_CAPPED
            return
        if not params.ignore_eos and seq.eos_token_id is not None and last == seq.eos_token_id:
            seq.status = SequenceStatus.FINISHED_STOPPED

    def _process_model_outputs(self, output: SamplerOutput, scheduler_outputs: SchedulerOutputs,
                               groups: Optional[List[SequenceGroup]] = None, lane: int = 0) -> List[RequestOutput]:
        now = time.monotonic()        # intervals (TTFT, TPOT, e2e) are taken against the monotonic arrival_time
        wall = time.time()
        scheduler = self.schedulers[lane]
        scheduled = list(scheduler_outputs.scheduled_seq_groups) if groups is None else groups
        if self.cache_config.enable_prefix_caching and (scheduler_outputs.prompt_run or self._decode_kv_shareable):
            # the step has written the KV of every full block of these sequences: later prompts with the
            # same leading tokens need not compute them again (llm_engine.py:543-547).  With split pools a
            # decode step's KV exists on the token pool only: a prompt worker could not read it, so only
            # prompt steps mark their blocks.
            for group in scheduled:
                scheduler.mark_blocks_as_computed(group)
        for group, out in zip(scheduled, output):
            if group.sampling_params.best_of > 1:
                self._process_forking_group(group, out, scheduler)
                if group.is_finished():
                    self._multi_sample_requests.discard(group.request_id)
                group.maybe_set_first_token_time(wall)
                continue
            seq = group.get_seqs(status=SequenceStatus.RUNNING)[0]
            sample = out.samples[0]
            seq.append_token_id(sample.output_token, sample.logprobs)
            self._decode_sequence(seq, group.sampling_params)
            self._check_stop(seq, group.sampling_params)
            if seq.is_finished():
                scheduler.free_seq(seq)
            group.maybe_set_first_token_time(wall)
        scheduler.free_finished_seq_groups()
        request_outputs = [RequestOutput.from_seq_group(g) for g in scheduled]
        request_outputs += [RequestOutput.from_seq_group(g) for g in scheduler_outputs.ignored_seq_groups]
        if self.stat_logger is not None:
            self.stat_logger.log(self._get_stats(scheduler_outputs, scheduled, now))
        return request_outputs

    def _process_forking_group(self, group: SequenceGroup, out, scheduler: Scheduler) -> None:
        """A request with best_of > 1 (no beam search): vllm/engine/llm_engine.py:424-534.  The prompt step returns
        best_of samples of the prompt sequence: all but the last become forks of it (their blocks shared
        copy-on-write, Scheduler.fork_seq), the last continues the prompt sequence itself; from then on every
        running sequence of the group gets the one sample that names it as its parent."""
        params = group.sampling_params
        parents = group.get_seqs(status=SequenceStatus.RUNNING)
        by_parent = {parent.seq_id: [] for parent in parents}
        for sample in out.samples:
            by_parent[sample.parent_seq_id].append(sample)
        children = []                                        # (sequence, its parent)
        for parent in parents:
            samples = by_parent[parent.seq_id]
            if not samples:                                  # nothing continues this sequence (llm_engine.py:446-452)
                parent.status = SequenceStatus.FINISHED_ABORTED
                group.remove(parent.seq_id)
                scheduler.free_seq(parent)
                continue
            for sample in samples[:-1]:
                child = parent.fork(self.seq_counter)
                self.seq_counter += 1
                child.append_token_id(sample.output_token, sample.logprobs)
                children.append((child, parent))
            parent.append_token_id(samples[-1].output_token, samples[-1].logprobs)
            children.append((parent, parent))
        for seq, _ in children:
            self._decode_sequence(seq, params)
            self._check_stop(seq, params)
        for seq, parent in children:
            if seq is not parent:
                group.add(seq)
                if not seq.is_finished():
                    scheduler.fork_seq(parent, seq)
        for seq, parent in children:                         # parents last: their forks share their blocks
            if seq is parent and seq.is_finished():
                scheduler.free_seq(seq)

    def _step_split(self) -> List[RequestOutput]:
        """One decode step on the token pool and one prompt step on the prompt pool, in flight
        together.  Hand-off epochs are counted here, once per prompt step of a slot, and given to
        both pools, so a signal and the wait that belongs to it always agree."""
        metas_d, outs_d = self.scheduler.schedule_decodes()
        metas_p, outs_p = self.scheduler.schedule_prompts()
        prompt_kwargs = decode_kwargs = None
        if metas_p:
            for slot in outs_p.blocks_to_nw:
                self._slot_epoch[(0, slot)] = self._slot_epoch.get((0, slot), 0) + 1
            prompt_kwargs = dict(seq_group_metadata_list=metas_p, blocks_to_swap_in={}, blocks_to_swap_out={},
                                 blocks_to_copy={}, blocks_to_nw=outs_p.blocks_to_nw,
                                 nw_epochs={s: self._slot_epoch[(0, s)] for s in outs_p.blocks_to_nw})
        if not outs_d.is_empty():
            decode_kwargs = dict(seq_group_metadata_list=metas_d, blocks_to_swap_in=outs_d.blocks_to_swap_in,
                                 blocks_to_swap_out=outs_d.blocks_to_swap_out, blocks_to_copy=outs_d.blocks_to_copy,
                                 blocks_to_nw=outs_d.blocks_to_nw,
                                 nw_epochs={s: self._slot_epoch.get((0, s), 0) for s in outs_d.blocks_to_nw})
        prompt_out, decode_out = self.model_executor.execute_split(prompt_kwargs, decode_kwargs)
        outputs: List[RequestOutput] = []
        if decode_out is not None and metas_d:
            outputs += self._process_model_outputs(decode_out, outs_d)
        if prompt_out is not None or outs_p.ignored_seq_groups:
            outputs += self._process_model_outputs(prompt_out or SamplerOutput([]), outs_p)
            self.scheduler.prompt_step_done([m.request_id for m in metas_p])
        return outputs

    # ----------------------------------------------------------- pipelined steps
    def _finished_by_length(self, seq: Sequence, params: SamplingParams) -> bool:
        """The stop rules of ``_check_stop`` that depend on lengths only: they are known as soon as
        a token has been sampled, before its value is."""
        return (seq.get_len() > self.scheduler_config.max_model_len
                or (params.max_tokens is not None and seq.get_output_len() == params.max_tokens))

    def _resolve_inflight(self) -> List[RequestOutput]:
        pending, scheduler_outputs, pairs = self._inflight
        self._inflight = None
        return self._apply_deferred(pending.resolve(), scheduler_outputs, pairs)

    def _apply_deferred(self, output: SamplerOutput, scheduler_outputs: SchedulerOutputs, pairs,
                        still_running=(), lane: int = 0) -> List[RequestOutput]:
        """Fills in the tokens of a step whose sequences have carried placeholders since its launch.
        ``still_running``: ids of the sequences in a newer step that is already in flight (split
        pools only): their blocks must outlive that step, because the prompt GPU does not queue
        behind the token GPU's stream when it writes new KV into reused blocks."""
        now = time.monotonic()        # intervals (TTFT, TPOT, e2e) are taken against the monotonic arrival_time
        wall = time.time()
        scheduler = self.schedulers[lane]
        for seq in self._free_later[lane]:               # their speculative extra step has resolved by now
            scheduler.free_seq(seq)
        self._free_later[lane] = []
        reported: List[SequenceGroup] = []
        if self.cache_config.enable_prefix_caching and self._decode_kv_shareable:
            # (split pools resolve only decode steps here, and those mark nothing: see _process_model_outputs)
            for group, seq in pairs:
                if not seq.is_finished() or seq.has_placeholder():
                    scheduler.mark_blocks_as_computed(group)
        for (group, seq), out in zip(pairs, output):
            if not seq.has_placeholder():
                continue        # ended (stop token, abort) before this step's token was accounted for: dropped
            sample = out.samples[0]
            seq.fill_placeholder(sample.output_token, sample.logprobs)
            if seq.status == SequenceStatus.FINISHED_ABORTED:
                continue
            self._decode_sequence(seq, group.sampling_params)
            if seq.is_finished():
                # ended by length when the step was launched; a stop string or token takes precedence as in
                # _check_stop
                if (self._stop_string_hit(seq, group.sampling_params)
                        or sample.output_token in group.sampling_params.stop_token_ids):
                    seq.status = SequenceStatus.FINISHED_STOPPED
                scheduler.free_seq(seq)            # no-op when the blocks went back at launch
            else:
                self._check_stop(seq, group.sampling_params)
                if seq.is_finished():
                    if seq.seq_id in still_running:
                        self._free_later[lane].append(seq)
                    else:
                        scheduler.free_seq(seq)
                    if group in scheduler.waiting:      # preempted for recomputation in the meantime
                        scheduler.waiting.remove(group)
            group.maybe_set_first_token_time(wall)
            reported.append(group)
        scheduler.free_finished_seq_groups()
        if self.stat_logger is not None:
            self.stat_logger.log(self._get_stats(scheduler_outputs, reported, now))
        return [RequestOutput.from_seq_group(g) for g in reported]

    def _launch_bookkeeping(self, pairs, lane: int = 0) -> None:
        """After a deferred step has been launched and the previous one resolved: every sequence of
        the new step gets its placeholder; the ones that end by length are retired at once."""
        scheduler = self.schedulers[lane]
        for group, seq in pairs:
            if seq.is_finished():
                continue            # a stop token of the previous step ended it; this step's sample is dropped
            seq.append_placeholder()
            if self._finished_by_length(seq, group.sampling_params):
                seq.status = SequenceStatus.FINISHED_LENGTH_CAPPED
                # co-located: device work is stream-ordered, so the blocks can be handed out again at
                # once.  Split pools: the prompt GPU writes new KV into the token GPU's blocks on its own
                # clock, so these blocks stay taken until the step that still reads them has resolved.
                if not self.pipeline_split:
                    scheduler.free_seq(seq)
        scheduler.free_finished_seq_groups()

    def _stage_prompt_step(self, lane: int, metas_p: List[SequenceGroupMetadata], outs_p: SchedulerOutputs):
        """The prompt pool keeps only the prompts in flight, under block numbers of its own: every prompt of
        the step gets consecutive blocks 0, 1, 2, ... of the half of the prompt cache that the previous step
        does not occupy, the commands carry those numbers, and the hand-off runs say where each of them goes
        in the token pool's cache.  -> (commands for the prompt pool, {slot: [(src, dst, length), ...]})"""
        half = self._staging_blocks // 2
        base = half * (self._staging_parity & 1)
        self._staging_parity += 1
        scheduler = self.schedulers[lane]
        staged: List[SequenceGroupMetadata] = []
        runs: Dict[int, List[tuple]] = {}
        cursor = 0
        for meta in metas_p:
            tables = {}
            for seq_id, table in meta.block_tables.items():
                if len(set(table)) != len(table):
                    raise NotImplementedError("a staged prompt cache cannot hold a sliding window's repeated blocks")
                mine = list(range(base + cursor, base + cursor + len(table)))
                cursor += len(table)
                tables[seq_id] = mine
                slot = scheduler.seq_to_slot_mapper.get_slot_id(seq_id)
                if slot in outs_p.blocks_to_nw:                  # due for a hand-off (every admitted prompt is)
                    runs.setdefault(slot, []).extend(coalesce_block_pairs(mine, table))
            staged.append(SequenceGroupMetadata(request_id=meta.request_id, is_prompt=True, seq_data=meta.seq_data,
                                                sampling_params=meta.sampling_params, block_tables=tables,
                                                computed_block_nums=[], state=meta.state))
        if cursor > half:
            raise RuntimeError(f"a prompt step of {cursor} blocks does not fit the prompt pool's {half}-block stage")
        return staged, runs

    _staging_parity = 0

    def _step_split_pipelined(self) -> List[RequestOutput]:
        """Prompt pool and token pool(s) run free of each other: a prompt step is started whenever the
        prompt pool is idle and collected when it has finished; every call starts the next decode
        step of every token pool and collects the previous one, so a token GPU always has a step queued.
        With several token pools a prompt step serves the waiting prompts of ONE pool (each pool has its own
        block space and hand-off slots); the pools take turns."""
        ex = self.model_executor
        lanes = len(self.schedulers)
        outputs: List[RequestOutput] = []
        # ---- prompt pool
        progressed = self._collect_finished_prompt_steps(outputs)
        if len(self._prompt_queue) < self._prompt_queue_depth:
            for turn in range(lanes):
                lane = (self._next_prompt_lane + turn) % lanes
                metas_p, outs_p = self.schedulers[lane].schedule_prompts()
                if metas_p:
                    break
                outputs += [RequestOutput.from_seq_group(g) for g in outs_p.ignored_seq_groups]
            if metas_p:                                        # (its ignored groups are reported when it resolves)
                self._next_prompt_lane = (lane + 1) % lanes
                nw = outs_p.blocks_to_nw
                commands = metas_p
                if self.staged_prompt_cache:
                    commands, nw = self._stage_prompt_step(lane, metas_p, outs_p)
                for slot in outs_p.blocks_to_nw:
                    self._slot_epoch[(lane, slot)] = self._slot_epoch.get((lane, slot), 0) + 1
                kwargs = dict(seq_group_metadata_list=commands, blocks_to_swap_in={}, blocks_to_swap_out={},
                              blocks_to_copy={}, blocks_to_nw=nw,
                              nw_epochs={s: self._slot_epoch[(lane, s)] for s in outs_p.blocks_to_nw})
                pending = ex.launch_prompt(kwargs, lane) if lane else ex.launch_prompt(kwargs)
                outs_p.scheduled_seq_groups = list(outs_p.scheduled_seq_groups)
                self._prompt_queue.append((pending, outs_p, [m.request_id for m in metas_p], lane))
                progressed = True
        # ---- token pools: start every pool's next step, then collect every pool's previous one
        started: List[Optional[tuple]] = [None] * lanes
        asked = [False] * lanes
        for lane in range(lanes):
            metas_d, outs_d = self.schedulers[lane].schedule_decodes()
            decode_kwargs = None
            if metas_d:
                pairs = [(g, g.get_seqs(status=SequenceStatus.RUNNING)[0]) for g in outs_d.scheduled_seq_groups]
                decode_kwargs = dict(seq_group_metadata_list=metas_d, blocks_to_swap_in=outs_d.blocks_to_swap_in,
                                     blocks_to_swap_out=outs_d.blocks_to_swap_out, blocks_to_copy=outs_d.blocks_to_copy,
                                     blocks_to_nw=outs_d.blocks_to_nw,
                                     nw_epochs={s: self._slot_epoch.get((lane, s), 0) for s in outs_d.blocks_to_nw})
                started[lane] = (outs_d, pairs)
            if decode_kwargs is not None or self._decode_inflight[lane] is not None:
                if lanes == 1:
                    ex.decode_send(decode_kwargs)
                else:
                    ex.decode_send(decode_kwargs, lane)
                asked[lane] = True
        for lane in range(lanes):
            if not asked[lane]:
                continue
            previous = ex.decode_collect() if lanes == 1 else ex.decode_collect(lane)
            if self._decode_inflight[lane] is not None:
                outs_prev, pairs_prev = self._decode_inflight[lane]
                self._decode_inflight[lane] = None
                newer = {seq.seq_id for _, seq in started[lane][1]} if started[lane] is not None else ()
                outputs += self._apply_deferred(previous, outs_prev, pairs_prev, still_running=newer, lane=lane)
            if started[lane] is not None:
                self._launch_bookkeeping(started[lane][1], lane)
                self._decode_inflight[lane] = started[lane]
            progressed = True
        if not progressed and self._prompt_queue:
            self._prompt_queue[0][0].wait()                    # nothing else to do: wait for the oldest prompt step
        # a prompt step that finished while this call sat in the token pools' collect is reported now, not a whole
        # decode step later: its first tokens are that much earlier with the caller (its sequences join the decode
        # batch at the next call either way)
        self._collect_finished_prompt_steps(outputs)
        self.scheduler.release_held()                          # aborted sequences whose last step has resolved
        return outputs

    def _collect_finished_prompt_steps(self, outputs: List[RequestOutput]) -> bool:
        """Resolves the prompt steps at the head of the queue that have finished on the device, in launch order;
        appends their requests' outputs.  -> whether there was one."""
        found = False
        while self._prompt_queue and self._prompt_queue[0][0].done():
            pending, outs_p, ids, lane = self._prompt_queue.popleft()
            sampled = pending.resolve()
            live = [(g, o) for g, o in zip(outs_p.scheduled_seq_groups, sampled)
                    if g.get_seqs(status=SequenceStatus.RUNNING)]              # not aborted meanwhile
            outputs += self._process_model_outputs(SamplerOutput([o for _, o in live]), outs_p,
                                                   groups=[g for g, _ in live], lane=lane)
            self.schedulers[lane].prompt_step_done(ids)
            found = True
        return found

    def _step_pipelined(self) -> List[RequestOutput]:
        metas, scheduler_outputs = self.scheduler.schedule()
        self.last_scheduled_prompt_run = bool(scheduler_outputs.prompt_run) and not scheduler_outputs.is_empty()
        launched = None
        if not scheduler_outputs.is_empty():
            pairs = [(g, g.get_seqs(status=SequenceStatus.RUNNING)[0])
                     for g in scheduler_outputs.scheduled_seq_groups]
            pending = self.model_executor.execute_model(
                metas, scheduler_outputs.blocks_to_swap_in, scheduler_outputs.blocks_to_swap_out,
                scheduler_outputs.blocks_to_copy, scheduler_outputs.blocks_to_nw, defer=True)
            launched = (pending, scheduler_outputs, pairs)
        # the previous step has finished by now (or does so while the new one runs)
        outputs = self._resolve_inflight() if self._inflight is not None else []
        if launched is not None:
            self._launch_bookkeeping(launched[2])
            self._inflight = launched
        outputs += [RequestOutput.from_seq_group(g) for g in scheduler_outputs.ignored_seq_groups]
        return outputs

    def step(self) -> List[RequestOutput]:
        """One scheduling decision, one model step on the executor, outputs appended
        (vllm/engine/llm_engine.py:571-633)."""
        if self.pipeline_split:
            return self._step_split_pipelined()
        if self.overlap_pools:
            return self._step_split()
        if self.pipeline_steps:
            if not self._multi_sample_requests:
                return self._step_pipelined()
            if self._inflight is not None:                   # finish the step in flight, then go on step by step
                return self._resolve_inflight()
        metas, scheduler_outputs = self.scheduler.schedule()
        self.last_scheduled_prompt_run = bool(scheduler_outputs.prompt_run) and not scheduler_outputs.is_empty()
        if not scheduler_outputs.is_empty():
            output = self.model_executor.execute_model(
                metas, scheduler_outputs.blocks_to_swap_in, scheduler_outputs.blocks_to_swap_out,
                scheduler_outputs.blocks_to_copy, scheduler_outputs.blocks_to_nw)
        else:
            output = SamplerOutput([])
        return self._process_model_outputs(output, scheduler_outputs)

    def _get_stats(self, scheduler_outputs: SchedulerOutputs, scheduled: List[SequenceGroup], now: float) -> Stats:
        total = self.cache_config.num_gpu_blocks or 1
        free = self.scheduler.block_manager.get_num_free_gpu_blocks()
        total_cpu = self.cache_config.num_cpu_blocks or 0
        cpu_usage = 1.0 - self.scheduler.block_manager.get_num_free_cpu_blocks() / total_cpu if total_cpu > 0 else 0.0
        prompt_tokens = scheduler_outputs.num_batched_tokens if scheduler_outputs.prompt_run else 0
        gen_tokens = 0 if scheduler_outputs.prompt_run else scheduler_outputs.num_batched_tokens
        ttft, tpot, e2e = [], [], []
        for g in scheduled:
            latency = g.get_last_latency(now)
            (ttft if scheduler_outputs.prompt_run else tpot).append(latency)
            if g.is_finished():
                e2e.append(now - g.metrics.arrival_time)
        return Stats(now=now, num_running=len(self.scheduler.running), num_swapped=len(self.scheduler.swapped),
                     num_waiting=len(self.scheduler.waiting), gpu_cache_usage=1.0 - free / total,
                     cpu_cache_usage=cpu_usage, num_prompt_tokens=prompt_tokens, num_generation_tokens=gen_tokens,
                     time_to_first_tokens=ttft, time_per_output_tokens=tpot, time_e2e_requests=e2e)

    def check_health(self) -> None:
        self.model_executor.check_health()
