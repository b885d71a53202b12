"""Iteration-level scheduler with Splitwise block tracking.

Decisions restated from vllm/core/scheduler.py:173-432: each step is either a PROMPT
run (admit waiting requests in arrival order while blocks, the token budget, the
sequence budget and the padding budget allow) or, if nothing could be admitted, a
DECODE run over everything running (preempting the youngest requests when blocks run
out, swapping requests back in when they fit).  With ``track_prompt_blocks`` every
sequence that has produced at most one token contributes its block table to
``blocks_to_nw[slot]`` (:269-277, :376-385), coalesced into (start, length) runs: the
prompt pool pushes these blocks, the token pool waits for them.

``schedule()`` reproduces the reference's serial alternation.  ``schedule_prompts()``
and ``schedule_decodes()`` expose the two halves separately so that a split
deployment can keep a prompt step and a decode step in flight at the same time
(the reference never overlaps the pools: vllm/executor/ray_gpu_executor.py:292-293).
"""
import enum
import time
from collections import defaultdict, deque
from typing import Callable, Deque, Dict, Iterable, List, Optional, Set, Tuple, Union

from ..config import CacheConfig, SchedulerConfig
from ..sequence import (Sequence, SequenceData, SequenceGroup, SequenceGroupMetadata, SequenceStatus)
from ..utils import SeqToSlotMapper, coalesce_blocks_by_id
from .block_manager import AllocStatus, BlockSpaceManager


class PreemptionMode(enum.Enum):
    SWAP = enum.auto()
    RECOMPUTE = enum.auto()


class SchedulerOutputs:
    """vllm/core/scheduler.py:32-80."""

    def __init__(self, scheduled_seq_groups: Iterable[SequenceGroup], prompt_run: bool, num_batched_tokens: int,
                 blocks_to_swap_in: Dict[int, int], blocks_to_swap_out: Dict[int, int],
                 blocks_to_copy: Dict[int, List[int]], blocks_to_nw: Dict[int, List[Tuple[int, int]]],
                 ignored_seq_groups: List[SequenceGroup]) -> None:
        # The reference orders every non-empty batch by (lora id, request id): its set of LoRA
        # requests is {None}, which has length 1, so _sort_by_lora_ids always runs
        # (vllm/core/scheduler.py:56-73).  Request ids compare as strings ('10' < '9').
        self.scheduled_seq_groups = sorted(scheduled_seq_groups, key=lambda g: g.request_id)
        self.prompt_run = prompt_run
        self.num_batched_tokens = num_batched_tokens
        self.blocks_to_swap_in = blocks_to_swap_in
        self.blocks_to_swap_out = blocks_to_swap_out
        self.blocks_to_copy = blocks_to_copy
        self.blocks_to_nw = blocks_to_nw
        assert not (blocks_to_swap_in and blocks_to_swap_out)
        self.ignored_seq_groups = ignored_seq_groups

    def is_empty(self) -> bool:
        return (not self.scheduled_seq_groups and not self.blocks_to_swap_in and not self.blocks_to_swap_out
                and not self.blocks_to_copy and not self.blocks_to_nw)


def _fcfs_order(now: float, groups: Iterable[SequenceGroup]) -> Deque[SequenceGroup]:
    """Oldest arrival first; the sort is stable (vllm/core/policy.py:16-36)."""
    return deque(sorted(groups, key=lambda g: now - g.metrics.arrival_time, reverse=True))


class Scheduler:
    def __init__(self, scheduler_config: SchedulerConfig, cache_config: CacheConfig, lora_config=None,
                 track_prompt_blocks: bool = False, strict_reference: bool = False,
                 one_cache: bool = False) -> None:
        """``strict_reference`` reproduces two behaviours of the reference that the default mode
        repairs: (1) a sequence re-admitted after preemption-by-recompute draws a NEW hand-off
        slot and its old one is never returned (vllm/core/scheduler.py:446-451 calls set_seq
        again; :496-507 frees only the blocks); (2) such a sequence is re-prefilled but, having
        more than one output token, is left out of blocks_to_nw (:273, :381), so the token pool
        would decode from stale KV.  By default the slot is kept and the re-prefilled blocks
        are pushed again."""
        if lora_config is not None:
            raise NotImplementedError("LoRA is outside the scope of this build")
        self.scheduler_config = scheduler_config
        self.cache_config = cache_config
        self.track_prompt_blocks = track_prompt_blocks
        self.strict_reference = strict_reference
        # steps for which a sequence's blocks are still announced in blocks_to_nw:
        # 2 after every (re-)prefill = the prompt step and the first decode step
        self._kv_pending: Dict[int, int] = {}
        self.seq_to_slot_mapper: Optional[SeqToSlotMapper] = SeqToSlotMapper() if track_prompt_blocks else None
        self.prompt_limit = min(scheduler_config.max_model_len, scheduler_config.max_num_batched_tokens)
        self.block_manager = BlockSpaceManager(
            block_size=cache_config.block_size, num_gpu_blocks=cache_config.num_gpu_blocks,
            num_cpu_blocks=cache_config.num_cpu_blocks, sliding_window=cache_config.sliding_window,
            enable_caching=cache_config.enable_prefix_caching)
        # split pools: only blocks written by a PROMPT step exist on both pools (the decode steps' KV lives on
        # the token pool alone), so only those may be shared with later prompts (the reference has no such
        # guard: ``strict_reference`` keeps its behaviour).  ``one_cache``: the prompt side and the decode side are
        # one GPU with one cache, so decode-filled blocks are as good as prompt-written ones.
        self.block_manager.promote_decode_blocks = not track_prompt_blocks or strict_reference or one_cache
        self.waiting: Deque[SequenceGroup] = deque()
        self.running: Deque[SequenceGroup] = deque()
        self.swapped: Deque[SequenceGroup] = deque()
        # split mode only: groups whose prompt step is still executing on the prompt pool
        self.prefilling: Set[str] = set()
        # pipelined split pools: the engine names the sequences of steps that are still running on a
        # pool; an abort must not hand their blocks (or hand-off slot) to anyone else before that
        # step has resolved, because the prompt GPU writes and pushes KV on its own clock
        self.in_flight: Callable[[Sequence], bool] = lambda seq: False
        self.held_after_abort: List[Sequence] = []

    # -------------------------------------------------------------- queue admin
    def add_seq_group(self, seq_group: SequenceGroup) -> None:
        self.waiting.append(seq_group)

    def abort_seq_group(self, request_id: Union[str, Iterable[str]]) -> None:
        ids = {request_id} if isinstance(request_id, str) else set(request_id)
        for queue in (self.waiting, self.running, self.swapped):
            hit = [g for g in queue if g.request_id in ids]
            for group in hit:
                ids.discard(group.request_id)
                queue.remove(group)
                self.prefilling.discard(group.request_id)
                for seq in group.get_seqs():
                    if not seq.is_finished():
                        seq.status = SequenceStatus.FINISHED_ABORTED
                        if self.in_flight(seq):
                            self.held_after_abort.append(seq)
                        else:
                            self.free_seq(seq)

    def release_held(self) -> None:
        """Frees the aborted sequences whose last step has resolved by now."""
        still = [seq for seq in self.held_after_abort if self.in_flight(seq)]
        for seq in self.held_after_abort:
            if seq not in still:
                self.free_seq(seq)
        self.held_after_abort = still

    def has_unfinished_seqs(self) -> bool:
        return bool(self.waiting or self.running or self.swapped)

    def get_num_unfinished_seq_groups(self) -> int:
        return len(self.waiting) + len(self.running) + len(self.swapped)

    # ------------------------------------------------------------ prompt half
    def _note_blocks_to_nw(self, seq: Sequence, blocks_to_nw: Dict[int, List[int]]) -> None:
        if self.strict_reference:
            due = seq.get_output_len() <= 1
        else:
            left = self._kv_pending.get(seq.seq_id, 0)
            due = left > 0
            if due:
                self._kv_pending[seq.seq_id] = left - 1
        if due:
            slot = self.seq_to_slot_mapper.get_slot_id(seq.seq_id)
            blocks_to_nw[slot].extend(self.block_manager.get_block_table(seq))

    def _hold_prompts(self) -> bool:
        """Optional prefill batching (off by default; the reference admits a prompt the moment it
        fits): while sequences are decoding, waiting prompts are held until at least
        ``prefill_batch_tokens`` tokens of them have gathered or the oldest has waited
        ``prefill_max_wait`` seconds, so that a prompt step runs GEMMs of several prompts' rows.
        The wait shrinks with the load: a prompt step stalls every running sequence for its whole
        length, so batching pays in proportion to how many are running (full wait from 64 up);
        with few running it would only add to the time to the first token."""
        cfg = self.scheduler_config
        want = getattr(cfg, "prefill_batch_tokens", 0)
        if want <= 0 or not self.waiting or not self.running:
            return False
        waited = time.monotonic() - self.waiting[0].metrics.arrival_time
        running = sum(group.num_seqs(status=SequenceStatus.RUNNING) for group in self.running)
        if waited < 0 or waited >= cfg.prefill_max_wait * min(1.0, running / 64.0):
            return False
        have = 0
        for group in self.waiting:
            have += group.get_seqs()[0].get_len()
            if have >= want:
                return False
        return True

    def _admit_prompts(self, blocks_to_nw: Dict[int, List[int]]):
        """-> (scheduled groups, ignored groups, padded token count)."""
        cfg = self.scheduler_config
        scheduled: List[SequenceGroup] = []
        ignored: List[SequenceGroup] = []
        if self._hold_prompts():
            return scheduled, ignored, 0
        num_curr_seqs = sum(g.get_max_num_running_seqs() for g in self.running)
        num_prompts, longest = 0, 0
        total_len = 0
        while self.waiting:
            group = self.waiting[0]
            waiting_seqs = group.get_seqs(status=SequenceStatus.WAITING)
            assert len(waiting_seqs) == 1, "Waiting sequence group should have only one prompt sequence."
            prompt_len = waiting_seqs[0].get_len()
            too_long = prompt_len > self.prompt_limit
            if not too_long:
                verdict = self.block_manager.can_allocate(group)
                if verdict == AllocStatus.LATER:
                    break
                too_long = verdict == AllocStatus.NEVER
            if too_long:
                for seq in waiting_seqs:
                    seq.status = SequenceStatus.FINISHED_IGNORED
                ignored.append(group)
                self.waiting.popleft()
                continue
            # the batch is padded to its longest prompt: budget the padded size
            padded = (num_prompts + 1) * max(longest, prompt_len)
            if padded > cfg.max_num_batched_tokens:
                break
            new_seqs = group.get_max_num_running_seqs()
            if num_curr_seqs + new_seqs > cfg.max_num_seqs:
                break
            if padded - (total_len + prompt_len) > cfg.max_paddings:
                break
            # every admitted sequence takes a hand-off slot, and a slot can outlive its sequence's place in ``running``
            # (a pipelined engine releases a finished sequence only when the step that still reads its blocks has
            # resolved): no free slot means "later", where the reference raises (vllm/utils.py:141-143)
            if self.seq_to_slot_mapper is not None:
                need = sum(1 for seq in waiting_seqs
                           if self.strict_reference or seq.seq_id not in self.seq_to_slot_mapper)
                if need > self.seq_to_slot_mapper.num_free():
                    break
            num_prompts += 1
            longest = max(longest, prompt_len)
            total_len += prompt_len
            self.waiting.popleft()
            self._allocate(group)
            self.running.append(group)
            num_curr_seqs += new_seqs
            scheduled.append(group)
            if self.track_prompt_blocks:
                for seq in group.get_seqs():
                    self._note_blocks_to_nw(seq, blocks_to_nw)
        return scheduled, ignored, num_prompts * longest

    # ------------------------------------------------------------ decode half
    def _reserve_decode_slots(self, now: float, blocks_to_swap_in, blocks_to_swap_out, blocks_to_copy,
                              skip: Optional[Set[str]] = None) -> None:
        """Gives every running group one more token slot, preempting from the back of the
        FCFS order when blocks run out; then swaps groups back in while they fit."""
        self.running = _fcfs_order(now, self.running)
        kept: Deque[SequenceGroup] = deque()
        preempted: List[SequenceGroup] = []
        while self.running:
            group = self.running.popleft()
            if skip and group.request_id in skip:
                kept.append(group)               # its prompt step is still in flight: no new slot yet
                continue
            while not self.block_manager.can_append_slot(group):
                victim = None
                for cand in reversed(self.running):
                    if not (skip and cand.request_id in skip):
                        victim = cand
                        break
                if victim is not None:
                    self.running.remove(victim)
                    self._preempt(victim, blocks_to_swap_out)
                    preempted.append(victim)
                else:
                    self._preempt(group, blocks_to_swap_out)
                    preempted.append(group)
                    break
            else:
                self._append_slot(group, blocks_to_copy)
                kept.append(group)
        self.running = kept

        self.swapped = _fcfs_order(now, self.swapped)
        if not preempted:
            num_curr_seqs = sum(g.get_max_num_running_seqs() for g in self.running)
            while self.swapped:
                group = self.swapped[0]
                if not self.block_manager.can_swap_in(group):
                    break
                new_seqs = group.get_max_num_running_seqs()
                if num_curr_seqs + new_seqs > self.scheduler_config.max_num_seqs:
                    break
                self.swapped.popleft()
                self._swap_in(group, blocks_to_swap_in)
                self._append_slot(group, blocks_to_copy)
                num_curr_seqs += new_seqs
                self.running.append(group)

    # --------------------------------------------------------------- schedule
    def _schedule(self) -> SchedulerOutputs:
        blocks_to_swap_in: Dict[int, int] = {}
        blocks_to_swap_out: Dict[int, int] = {}
        blocks_to_copy: Dict[int, List[int]] = {}
        blocks_to_nw: Dict[int, List[int]] = defaultdict(list)
        now = time.monotonic()

        if not self.swapped:
            scheduled, ignored, padded_tokens = self._admit_prompts(blocks_to_nw)
            if scheduled or ignored:
                return SchedulerOutputs(scheduled, True, padded_tokens, blocks_to_swap_in, blocks_to_swap_out,
                                        blocks_to_copy, coalesce_blocks_by_id(blocks_to_nw), ignored)

        self._reserve_decode_slots(now, blocks_to_swap_in, blocks_to_swap_out, blocks_to_copy)
        num_batched_tokens = sum(g.num_seqs(status=SequenceStatus.RUNNING) for g in self.running)
        if self.track_prompt_blocks:
            for group in self.running:
                for seq in group.get_seqs(status=SequenceStatus.RUNNING):
                    self._note_blocks_to_nw(seq, blocks_to_nw)
        return SchedulerOutputs(self.running, False, num_batched_tokens, blocks_to_swap_in, blocks_to_swap_out,
                                blocks_to_copy, coalesce_blocks_by_id(blocks_to_nw), [])

    def _metadata(self, outputs: SchedulerOutputs, groups: Iterable[SequenceGroup]) -> List[SequenceGroupMetadata]:
        now = time.monotonic()        # the clock of arrival_time: time_in_queue is a difference of the two
        metas: List[SequenceGroupMetadata] = []
        for group in groups:
            group.maybe_set_first_scheduled_time(now)
            seq_data: Dict[int, SequenceData] = {}
            block_tables: Dict[int, List[int]] = {}
            for seq in group.get_seqs(status=SequenceStatus.RUNNING):
                seq_data[seq.seq_id] = seq.data
                block_tables[seq.seq_id] = self.block_manager.get_block_table(seq)
                self.block_manager.access_all_blocks_in_seq(seq, now)     # the evictor's clock (scheduler.py:418)
            metas.append(SequenceGroupMetadata(
                request_id=group.request_id, is_prompt=outputs.prompt_run, seq_data=seq_data,
                sampling_params=group.sampling_params, block_tables=block_tables,
                # leading blocks whose KV is in the cache already: the prompt step skips them (scheduler.py:427-428)
                computed_block_nums=self.block_manager.get_common_computed_block_ids(group), state=group.state))
        return metas

    def mark_blocks_as_computed(self, seq_group: SequenceGroup) -> None:
        """vllm/core/scheduler.py:543-544."""
        self.block_manager.mark_blocks_as_computed(seq_group)

    def schedule(self) -> Tuple[List[SequenceGroupMetadata], SchedulerOutputs]:
        outputs = self._schedule()
        return self._metadata(outputs, outputs.scheduled_seq_groups), outputs

    # ------------------------------------------------------------- split mode
    def schedule_prompts(self) -> Tuple[List[SequenceGroupMetadata], SchedulerOutputs]:
        """One prompt step for the prompt pool; admitted groups join ``running`` but are held out
        of decode steps until ``prompt_step_done`` reports their first token."""
        blocks_to_nw: Dict[int, List[int]] = defaultdict(list)
        scheduled, ignored = [], []
        padded = 0
        if not self.swapped:
            scheduled, ignored, padded = self._admit_prompts(blocks_to_nw)
        for group in scheduled:
            self.prefilling.add(group.request_id)
        outputs = SchedulerOutputs(scheduled, True, padded, {}, {}, {}, coalesce_blocks_by_id(blocks_to_nw),
                                   ignored)
        return self._metadata(outputs, outputs.scheduled_seq_groups), outputs

    def prompt_step_done(self, request_ids: Iterable[str]) -> None:
        for rid in request_ids:
            self.prefilling.discard(rid)

    def schedule_decodes(self) -> Tuple[List[SequenceGroupMetadata], SchedulerOutputs]:
        """One decode step for the token pool over the running groups whose prompt has finished."""
        blocks_to_swap_in: Dict[int, int] = {}
        blocks_to_swap_out: Dict[int, int] = {}
        blocks_to_copy: Dict[int, List[int]] = {}
        blocks_to_nw: Dict[int, List[int]] = defaultdict(list)
        self._reserve_decode_slots(time.monotonic(), blocks_to_swap_in, blocks_to_swap_out, blocks_to_copy,
                                   skip=self.prefilling)
        ready = [g for g in self.running if g.request_id not in self.prefilling]
        if self.track_prompt_blocks:
            for group in ready:
                for seq in group.get_seqs(status=SequenceStatus.RUNNING):
                    self._note_blocks_to_nw(seq, blocks_to_nw)
        tokens = sum(g.num_seqs(status=SequenceStatus.RUNNING) for g in ready)
        outputs = SchedulerOutputs(ready, False, tokens, blocks_to_swap_in, blocks_to_swap_out, blocks_to_copy,
                                   coalesce_blocks_by_id(blocks_to_nw), [])
        return self._metadata(outputs, outputs.scheduled_seq_groups), outputs

    # -------------------------------------------------------------- bookkeeping
    def fork_seq(self, parent_seq: Sequence, child_seq: Sequence) -> None:
        self.block_manager.fork(parent_seq, child_seq)

    def free_seq(self, seq: Sequence) -> None:
        self.block_manager.free(seq)
        if self.seq_to_slot_mapper is not None and seq.seq_id in self.seq_to_slot_mapper:
            self.seq_to_slot_mapper.free_seq(seq.seq_id)
        self._kv_pending.pop(seq.seq_id, None)

    def free_finished_seq_groups(self) -> None:
        self.running = deque(g for g in self.running if not g.is_finished())

    def _allocate(self, seq_group: SequenceGroup) -> None:
        self.block_manager.allocate(seq_group)
        for seq in seq_group.get_seqs(status=SequenceStatus.WAITING):
            seq.status = SequenceStatus.RUNNING
            if self.seq_to_slot_mapper is not None:
                if self.strict_reference or seq.seq_id not in self.seq_to_slot_mapper:
                    self.seq_to_slot_mapper.set_seq(seq.seq_id)
                self._kv_pending[seq.seq_id] = 2

    def _append_slot(self, seq_group: SequenceGroup, blocks_to_copy: Dict[int, List[int]]) -> None:
        for seq in seq_group.get_seqs(status=SequenceStatus.RUNNING):
            cow = self.block_manager.append_slot(seq)
            if cow is not None:
                src, dst = cow
                blocks_to_copy.setdefault(src, []).append(dst)

    def _preempt(self, seq_group: SequenceGroup, blocks_to_swap_out: Dict[int, int],
                 preemption_mode: Optional[PreemptionMode] = None) -> None:
        # a lone sequence is recomputed (cheaper); groups with several sequences are swapped out
        # (vllm/core/scheduler.py:467-494)
        self.num_preemptions = getattr(self, "num_preemptions", 0) + 1
        if preemption_mode is None:
            preemption_mode = (PreemptionMode.RECOMPUTE if seq_group.get_max_num_running_seqs() == 1
                               else PreemptionMode.SWAP)
        if preemption_mode == PreemptionMode.RECOMPUTE:
            seqs = seq_group.get_seqs(status=SequenceStatus.RUNNING)
            assert len(seqs) == 1
            for seq in seqs:
                seq.status = SequenceStatus.WAITING
                self.block_manager.free(seq)
            self.waiting.appendleft(seq_group)
        else:
            if not self.block_manager.can_swap_out(seq_group):
                raise RuntimeError("Aborted due to the lack of CPU swap space. Please increase the swap space "
                                   "to avoid this error.")
            blocks_to_swap_out.update(self.block_manager.swap_out(seq_group))
            for seq in seq_group.get_seqs(status=SequenceStatus.RUNNING):
                seq.status = SequenceStatus.SWAPPED
            self.swapped.append(seq_group)

    def _swap_in(self, seq_group: SequenceGroup, blocks_to_swap_in: Dict[int, int]) -> None:
        blocks_to_swap_in.update(self.block_manager.swap_in(seq_group))
        for seq in seq_group.get_seqs(status=SequenceStatus.SWAPPED):
            seq.status = SequenceStatus.RUNNING


class _PooledBlockCounts:
    """Free-block totals over several block managers (statistics only)."""

    def __init__(self, managers: List[BlockSpaceManager]) -> None:
        self.managers = managers

    def get_num_free_gpu_blocks(self) -> int:
        return sum(m.get_num_free_gpu_blocks() for m in self.managers)

    def get_num_free_cpu_blocks(self) -> int:
        return sum(m.get_num_free_cpu_blocks() for m in self.managers)


class SchedulerGroup:
    """The schedulers of several token pools behind the queue-administration interface of one.

    Each token pool is a block space and a set of hand-off slots of its own, so each has its own
    ``Scheduler``; a request is bound to a pool when it arrives -- the one with the fewest unfinished
    requests -- and stays there (its KV lives in that pool's cache).  The reference's world has a single
    token pool (vllm/config.py:434-438) and needs none of this.  Scheduling itself is asked of the
    member schedulers directly (``LLMEngine.schedulers``)."""

    def __init__(self, lanes: List[Scheduler]) -> None:
        self.lanes = lanes
        self.block_manager = _PooledBlockCounts([s.block_manager for s in lanes])

    def add_seq_group(self, seq_group: SequenceGroup) -> None:
        lane = min(range(len(self.lanes)), key=lambda i: self.lanes[i].get_num_unfinished_seq_groups())
        self.lanes[lane].add_seq_group(seq_group)

    def abort_seq_group(self, request_id: Union[str, Iterable[str]]) -> None:
        ids = [request_id] if isinstance(request_id, str) else list(request_id)
        for lane in self.lanes:
            lane.abort_seq_group(ids)

    def release_held(self) -> None:
        for lane in self.lanes:
            lane.release_held()

    def has_unfinished_seqs(self) -> bool:
        return any(lane.has_unfinished_seqs() for lane in self.lanes)

    def get_num_unfinished_seq_groups(self) -> int:
        return sum(lane.get_num_unfinished_seq_groups() for lane in self.lanes)

    @property
    def waiting(self) -> List[SequenceGroup]:
        return [g for lane in self.lanes for g in lane.waiting]

    @property
    def running(self) -> List[SequenceGroup]:
        return [g for lane in self.lanes for g in lane.running]

    @property
    def swapped(self) -> List[SequenceGroup]:
        return [g for lane in self.lanes for g in lane.swapped]

    @property
    def held_after_abort(self) -> List[Sequence]:
        return [s for lane in self.lanes for s in lane.held_after_abort]

    @property
    def num_preemptions(self) -> int:
        return sum(getattr(lane, "num_preemptions", 0) for lane in self.lanes)
