"""Known-answer behaviour of the scheduler, of the sliding-window block accounting and of the small API
containers: the scenarios and expected answers of the reference's tests/core/test_scheduler.py:11-170
(admission, abort, prompt-then-decode scheduling, preemption by recomputation followed by an abort,
the running-sequence cap), tests/core/test_block_manager.py:279-363 (two sequences sharing a sliding
window), tests/test_sequence.py:21-50 (SamplerOutput as a container) and tests/test_sampling_params.py:6-8,
restated against this build's classes.  Both scheduler flavours are checked where the scenario does not
depend on the slot-keeping extension (``strict_reference`` False is this build's default)."""
import time

import pytest

import conftest  # noqa: F401
from splitwise_amd.config import CacheConfig, SchedulerConfig
from splitwise_amd.core.block_manager import BlockSpaceManager
from splitwise_amd.core.scheduler import Scheduler
from splitwise_amd.sampling_params import SamplingParams
from splitwise_amd.sequence import (Logprob, SamplerOutput, Sequence, SequenceGroup, SequenceGroupOutput,
                                    SequenceOutput)

BLOCK = 4


def make_scheduler(max_tokens, max_seqs, max_model_len, blocks, strict):
    cache = CacheConfig(BLOCK, 1.0, 1, "auto")
    cache.num_gpu_blocks = blocks
    cache.num_cpu_blocks = blocks
    return Scheduler(SchedulerConfig(max_tokens, max_seqs, max_model_len, 256), cache, None, strict_reference=strict)


def prompt(request_id, length=BLOCK):
    seq = Sequence(int(request_id), None, list(range(length)), BLOCK)
    return seq, SequenceGroup(request_id, [seq], SamplingParams(), time.time())


def no_block_moves(out):
    return not out.blocks_to_copy and not out.blocks_to_swap_in and not out.blocks_to_swap_out


@pytest.mark.parametrize("strict", [True, False])
def test_every_added_group_counts_as_unfinished(strict):
    sched = make_scheduler(100, 64, 1, 4, strict)
    for i in range(4):
        sched.add_seq_group(prompt(str(i))[1])
        assert sched.get_num_unfinished_seq_groups() == i + 1


@pytest.mark.parametrize("strict", [True, False])
def test_abort_takes_a_set_of_request_ids(strict):
    sched = make_scheduler(100, 64, 1, 4, strict)
    ids = {str(i) for i in range(4)}
    for i in sorted(ids):
        sched.add_seq_group(prompt(i)[1])
    assert sched.get_num_unfinished_seq_groups() == 4
    sched.abort_seq_group(ids)
    assert sched.get_num_unfinished_seq_groups() == 0


@pytest.mark.parametrize("strict", [True, False])
def test_prompts_go_in_one_step_and_decode_in_the_next(strict):
    sched = make_scheduler(64, 4, 16, 8, strict)
    groups = [prompt(str(i))[1] for i in range(4)]
    for g in groups:
        sched.add_seq_group(g)
    metas, out = sched.schedule()                      # the prompt run: all four, 4 tokens each
    assert set(out.scheduled_seq_groups) == set(groups)
    assert out.num_batched_tokens == 4 * BLOCK
    assert no_block_moves(out) and len(metas) == 4
    metas, out = sched.schedule()                      # one decode token each
    assert set(out.scheduled_seq_groups) == set(groups)
    assert out.num_batched_tokens == 4
    assert no_block_moves(out) and len(metas) == 4


@pytest.mark.parametrize("strict", [True, False])
def test_the_newest_group_is_preempted_for_recomputation_and_returns_after_an_abort(strict):
    # two blocks in all: both prompts fit, but only one of the sequences can grow into a second block
    sched = make_scheduler(64, 2, 16, 2, strict)
    seq_a, group_a = prompt("1")
    seq_b, group_b = prompt("2")
    sched.add_seq_group(group_a)
    sched.add_seq_group(group_b)
    metas, out = sched.schedule()
    assert out.scheduled_seq_groups == [group_a, group_b]
    assert out.num_batched_tokens == 2 * BLOCK and no_block_moves(out) and len(metas) == 2
    assert sched.get_num_unfinished_seq_groups() == 2
    for seq in (seq_a, seq_b):
        seq.append_token_id(0, {0: Logprob(0.0)})
    metas, out = sched.schedule()                      # b gives its block back, a decodes
    assert out.scheduled_seq_groups == [group_a]
    assert out.num_batched_tokens == 1 and no_block_moves(out) and len(metas) == 1
    assert sched.get_num_unfinished_seq_groups() == 2
    sched.abort_seq_group("1")
    metas, out = sched.schedule()                      # b's prompt (now 5 tokens) is computed again
    assert out.scheduled_seq_groups == [group_b]
    assert out.num_batched_tokens == seq_b.get_len() and no_block_moves(out) and len(metas) == 1
    assert sched.get_num_unfinished_seq_groups() == 1


@pytest.mark.parametrize("strict", [True, False])
def test_running_sequences_are_capped_at_max_num_seqs(strict):
    sched = make_scheduler(64, 2, 16, 8, strict)
    groups = [prompt(str(i))[1] for i in range(4)]
    sched.add_seq_group(groups[0])
    assert set(sched.schedule()[1].scheduled_seq_groups) == {groups[0]}      # its prompt
    assert set(sched.schedule()[1].scheduled_seq_groups) == {groups[0]}      # its first decode step
    sched.add_seq_group(groups[1])
    sched.add_seq_group(groups[2])
    # one sequence is running, the cap is two: only one of the two new prompts is admitted
    assert set(sched.schedule()[1].scheduled_seq_groups) == {groups[1]}


def test_two_sequences_sharing_a_sliding_window():
    window, blocks = 2, 8
    mgr = BlockSpaceManager(1, blocks, blocks, watermark=0, sliding_window=window)
    assert mgr.get_num_free_gpu_blocks() == blocks
    parent = Sequence(1, None, [0, 1, 2], 1)
    mgr.allocate(SequenceGroup("1", [parent], SamplingParams(), time.time()))
    assert mgr.get_num_free_gpu_blocks() == blocks - window          # three tokens, but the window holds two
    child = parent.fork(2)
    mgr.fork(parent, child)
    assert mgr.get_num_free_gpu_blocks() == blocks - window          # a fork costs nothing
    assert mgr.get_block_table(parent) == mgr.get_block_table(child)
    child.append_token_id(4, {4: Logprob(0.0)})
    mgr.append_slot(child)                                           # the shared block is copied on write
    assert mgr.get_num_free_gpu_blocks() == blocks - window - 1
    parent.append_token_id(5, {5: Logprob(0.0)})
    mgr.append_slot(parent)
    assert mgr.get_num_free_gpu_blocks() == blocks - window - 1      # still one block shared, one each of their own
    table_parent, table_child = mgr.get_block_table(parent), mgr.get_block_table(child)
    assert table_parent != table_child
    assert table_parent[-2] == table_child[-2]                       # the second-last block is the shared one
    mgr.free(parent)
    assert mgr.get_num_free_gpu_blocks() == blocks - window          # the child keeps a full window
    mgr.free(child)
    assert mgr.get_num_free_gpu_blocks() == blocks


def _group_outputs(n):
    return [SequenceGroupOutput(samples=[SequenceOutput(parent_seq_id=0, output_token=i, logprobs={})],
                                prompt_logprobs=None) for i in range(n)]


def test_sampler_output_is_a_list_like_container():
    items = _group_outputs(5)
    out = SamplerOutput(outputs=items)
    assert len(out) == 5
    assert out.sampled_token_probs is None and out.sampled_token_ids is None
    assert out.spec_decode_worker_metrics is None
    assert out[2] == items[2]
    replacement = SequenceGroupOutput(samples=[SequenceOutput(parent_seq_id=0, output_token=99, logprobs={})],
                                      prompt_logprobs=None)
    out[2] = replacement
    assert out[2] == replacement
    assert SamplerOutput(outputs=_group_outputs(5)) == SamplerOutput(outputs=list(_group_outputs(5)))
    assert SamplerOutput(outputs=_group_outputs(5)) != SamplerOutput(outputs=_group_outputs(4))


def test_sampling_params_accept_no_token_limit():
    SamplingParams(temperature=0.01, top_p=0.1, max_tokens=None)


def test_decode_kernel_choice_takes_raggedness_into_account():
    """v1 (one workgroup per sequence and head group) against v2 (512-token partitions): short contexts and large
    launches stay unpartitioned; a launch of a few hundred workgroups whose longest context is at least twice the
    mean goes partitioned when the mean is known (eager steps)."""
    from splitwise_amd.model_executor.layers.attention import PagedAttentionImpl
    use_v1 = PagedAttentionImpl.use_v1
    assert use_v1(256, 32, 32, 500)                                    # one partition: nothing to split
    assert use_v1(256, 32, 32, 8192)                                   # 8192 workgroups
    assert use_v1(256, 32, 32, 8192, 1, mean_context_len=1000.0)       # ... ragged or not
    assert use_v1(256, 16, 2, 8192, 8)                                 # 512 matrix-core workgroups, mean unknown
    assert not use_v1(256, 16, 2, 8192, 8, mean_context_len=1009.0)    # the same launch, known to be ragged
    assert use_v1(256, 16, 2, 8192, 8, mean_context_len=7000.0)        # long but even
    assert use_v1(64, 64, 8, 900, 8, mean_context_len=100.0)           # ragged but short: two partitions at most


def test_ragged_decode_batches_are_ordered_longest_first_and_mapped_back():
    from splitwise_amd.sequence import SequenceData, SequenceGroupMetadata
    from splitwise_amd.worker.model_runner import ModelRunner
    runner = ModelRunner.__new__(ModelRunner)            # only the ordering rule is exercised

    def metas(lens):
        return [SequenceGroupMetadata(str(i), False, {i: SequenceData(list(range(n)))}, SamplingParams(), {i: [0]})
                for i, n in enumerate(lens)]

    assert runner._longest_first(metas([100] * 32)) is None                     # even lengths: left alone
    assert runner._longest_first(metas([100] * 8 + [5000])) is None             # small batch: left alone
    lens = [50] * 20 + [4000, 60, 3000] + [40] * 9
    order, inverse = runner._longest_first(metas(lens))
    assert [lens[i] for i in order] == sorted(lens, reverse=True)
    assert order[:2] == [20, 22] and order[2] == 21                              # stable among equal lengths
    assert all(order[inverse[i]] == i for i in range(len(lens)))
    out = SamplerOutput(outputs=[SequenceGroupOutput([SequenceOutput(i, i, {})], None) for i in order])
    out.reorder(inverse)                                                         # back to the order asked for
    assert [g.samples[0].parent_seq_id for g in out] == list(range(len(lens)))
    assert runner._longest_first(metas(sorted(lens, reverse=True))) is None      # already longest first


def test_dense_copies_of_int4_weights_respect_the_budget(monkeypatch):
    import gc
    import torch
    from splitwise_amd.model_executor.layers.quantization import dense_copy
    monkeypatch.setattr(dense_copy, "_BUDGET_BYTES", dense_copy._used_bytes + 3000)
    made = []

    def reconstruct():
        made.append(1)
        return torch.zeros(32, 32, dtype=torch.float16)              # 2048 bytes

    layer_a, layer_b = {}, {}
    w = dense_copy.dense_weight(layer_a, 32, 32, reconstruct)
    assert w is not None and dense_copy.dense_weight(layer_a, 32, 32, reconstruct) is w and len(made) == 1
    assert dense_copy.dense_weight(layer_b, 32, 32, reconstruct) is None            # 4096 bytes would exceed the budget
    assert dense_copy.dense_weight(layer_b, 32, 32, reconstruct) is None and len(made) == 1
    used = dense_copy._used_bytes
    del w
    layer_a.clear()
    gc.collect()
    assert dense_copy._used_bytes == used - 2048                                     # given back with the layer


@pytest.mark.parametrize("split", [False, True])
def test_a_prompt_that_extends_an_earlier_output_shares_only_blocks_a_prompt_step_wrote(split):
    """Prefix caching, multi-turn: request B's prompt is request A's prompt plus A's generated text.  Co-located,
    every full block of A (prompt and decode) is in the one cache and may be reported as computed.  With split
    pools the blocks that filled up during A's decode steps hold KV on the token pool only: B's prompt worker
    must compute them itself, so they may be neither shared by content nor reported as computed."""
    cache = CacheConfig(BLOCK, 1.0, 1, "auto", enable_prefix_caching=True)
    cache.num_gpu_blocks = cache.num_cpu_blocks = 32
    sched = Scheduler(SchedulerConfig(64, 8, 64, 256), cache, None, track_prompt_blocks=split)
    seq_a = Sequence(0, None, list(range(100, 100 + 2 * BLOCK)), BLOCK)             # two full prompt blocks
    group_a = SequenceGroup("a", [seq_a], SamplingParams(), time.monotonic())
    sched.add_seq_group(group_a)

    def engine_marks(group, prompt_run):          # what LLMEngine._process_model_outputs does after a step
        if prompt_run or not split:
            sched.mark_blocks_as_computed(group)

    metas, out = (sched.schedule_prompts() if split else sched.schedule())
    assert out.prompt_run and len(metas) == 1
    engine_marks(group_a, True)
    if split:
        sched.prompt_step_done(["a"])
    generated = []
    for step in range(2 * BLOCK + 1):             # A's first token comes from the prompt step; then 2 blocks more
        tok = 500 + step
        generated.append(tok)
        seq_a.append_token_id(tok, {tok: Logprob(0.0)})
        metas, out = (sched.schedule_decodes() if split else sched.schedule())
        assert not out.prompt_run and len(metas) == 1
        engine_marks(group_a, False)
    table_a = sched.block_manager.get_block_table(seq_a)
    assert len(table_a) == 5
    # B = A's prompt + A's output so far + one new block of its own
    seq_b = Sequence(1, None, list(range(100, 100 + 2 * BLOCK)) + generated + [900 + i for i in range(BLOCK)], BLOCK)
    group_b = SequenceGroup("b", [seq_b], SamplingParams(), time.monotonic())
    sched.add_seq_group(group_b)
    metas, out = (sched.schedule_prompts() if split else sched.schedule())
    assert out.prompt_run and [m.request_id for m in metas] == ["b"]
    table_b = sched.block_manager.get_block_table(seq_b)
    computed = metas[0].computed_block_nums
    if split:
        # only A's prompt-step blocks are shared and computed: the first of its two prompt blocks is reported
        # (the reference never marks the last full block of a step), nothing written by a decode step is
        assert table_b[:2] == table_a[:2]
        assert not set(table_b[2:]) & set(table_a[2:])
        assert computed == table_a[:1]
    else:
        assert table_b[:4] == table_a[:4]
        assert computed == table_a[:len(computed)] and len(computed) >= 3


def test_prompt_admission_waits_for_a_free_hand_off_slot():
    """A hand-off slot can outlive its sequence's place in ``running`` (a pipelined split engine releases a finished
    sequence only when the step that still reads its blocks has resolved).  With every slot taken the next prompt
    waits; the reference's slot map raises "No more slots available" there (vllm/utils.py:141-143)."""
    from splitwise_amd.utils import SeqToSlotMapper
    cache = CacheConfig(BLOCK, 1.0, 1, "auto")
    cache.num_gpu_blocks = cache.num_cpu_blocks = 64
    sched = Scheduler(SchedulerConfig(64, 8, 64, 256), cache, None, track_prompt_blocks=True)
    sched.seq_to_slot_mapper = SeqToSlotMapper(num_slots=2)
    groups = []
    for i in range(3):
        seq = Sequence(i, None, [1, 2, 3, 4, 5], BLOCK)
        groups.append(SequenceGroup(str(i), [seq], SamplingParams(max_tokens=4), time.monotonic()))
        sched.add_seq_group(groups[-1])
    metas, out = sched.schedule_prompts()
    assert [m.request_id for m in metas] == ["0", "1"] and len(sched.waiting) == 1
    metas, out = sched.schedule_prompts()
    assert metas == [] and len(sched.waiting) == 1                  # no slot: later, not an exception
    # sequence 0 finishes; the engine drops it from ``running`` at once but frees it one step later
    seq0 = groups[0].get_seqs()[0]
    from splitwise_amd.sequence import SequenceStatus
    seq0.status = SequenceStatus.FINISHED_LENGTH_CAPPED
    sched.free_finished_seq_groups()
    assert sched.schedule_prompts()[0] == []                        # its slot is still taken
    sched.free_seq(seq0)
    metas, _ = sched.schedule_prompts()
    assert [m.request_id for m in metas] == ["2"]
    assert sched.seq_to_slot_mapper.get_slot_id(2) == 0             # the freed slot is the next one handed out
