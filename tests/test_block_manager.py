"""Known-answer behaviour of the block manager: the scenarios and expected answers of the
reference's tests/core/test_block_manager.py:59-276 (allocate with and without watermark,
append_slot, copy-on-write, fork, swap out/in, free, reset), restated against this build's
``BlockSpaceManager``, plus the block numbering rule (fresh ids 0,1,2,... then first-freed first,
block_manager.py:42-55, evictor.py:131-137)."""
import time

import pytest

import conftest  # noqa: F401
from splitwise_amd.core.block_manager import AllocStatus, BlockSpaceManager
from splitwise_amd.sampling_params import SamplingParams
from splitwise_amd.sequence import Logprob, Sequence, SequenceGroup, SequenceStatus

_next_id = [0]


def _prompt(request_id, prompt_length, block_size=4):
    _next_id[0] += 1
    seq = Sequence(_next_id[0], None, list(range(prompt_length)), block_size)
    return seq, SequenceGroup(request_id, [seq], SamplingParams(), time.time())


def _append(seq, token):
    seq.append_token_id(token, {token: Logprob(0.0)})


def test_allocate_fills_the_pool_and_respects_the_watermark():
    mgr = BlockSpaceManager(4, 4, 4, watermark=0)
    for i in range(4):
        _, group = _prompt(str(i), 4)
        assert mgr.can_allocate(group) == AllocStatus.OK
        mgr.allocate(group)
    assert mgr.can_allocate(group) != AllocStatus.OK
    mgr = BlockSpaceManager(4, 4, 4, watermark=1 / 4)             # one block held back
    for i in range(3):
        _, group = _prompt(str(i), 4)
        assert mgr.can_allocate(group) == AllocStatus.OK
        mgr.allocate(group)
    assert mgr.can_allocate(group) != AllocStatus.OK


def test_a_prompt_larger_than_the_pool_is_never_allocatable():
    mgr = BlockSpaceManager(4, 4, 4, watermark=0)
    _, group = _prompt("big", 17)
    assert mgr.can_allocate(group) == AllocStatus.NEVER


def test_append_slot_takes_a_block_only_at_a_block_boundary():
    mgr = BlockSpaceManager(4, 4, 4, watermark=0)
    seq, group = _prompt("1", 4)
    mgr.allocate(group)
    assert mgr.can_append_slot(group)
    before = mgr.get_num_free_gpu_blocks()
    assert not mgr.append_slot(seq)                               # no new logical block yet
    assert mgr.get_num_free_gpu_blocks() == before
    for i in range(4):
        _append(seq, i + 5)
    assert mgr.can_append_slot(group)
    assert not mgr.append_slot(seq)
    assert before - mgr.get_num_free_gpu_blocks() == 1


def test_append_slot_copies_on_write_when_the_last_block_is_shared():
    mgr = BlockSpaceManager(4, 4, 4, watermark=0)
    parent = Sequence(1, "one two three", [1, 2, 3], 4)
    child = parent.fork(2)
    group = SequenceGroup("1", [parent, child], SamplingParams(), time.time())
    mgr.allocate(group)
    _append(child, 4)
    mgr.fork(parent, child)
    assert mgr.can_append_slot(group)
    before = mgr.get_num_free_gpu_blocks()
    pair = mgr.append_slot(child)
    assert pair is not None and pair[0] != pair[1]
    assert before - mgr.get_num_free_gpu_blocks() == 1


def test_fork_shares_the_table_until_the_child_writes():
    mgr = BlockSpaceManager(4, 4, 4, watermark=0)
    parent, group = _prompt("1", 3)
    mgr.allocate(group)
    child = parent.fork(99)
    mgr.fork(parent, child)
    assert mgr.get_block_table(parent) == mgr.get_block_table(child)
    _append(child, 4)
    mgr.append_slot(child)
    assert mgr.get_block_table(parent) != mgr.get_block_table(child)


def test_swap_out_and_in_move_every_block_and_keep_the_accounting():
    mgr = BlockSpaceManager(4, 4, 4, watermark=0)
    seq, group = _prompt("1", 3)
    seq.status = SequenceStatus.WAITING
    mgr.allocate(group)
    seq.status = SequenceStatus.RUNNING
    _append(seq, 0)
    gpu_blocks = mgr.get_block_table(seq)
    assert mgr.can_swap_out(group)
    cpu_before, gpu_before = mgr.get_num_free_cpu_blocks(), mgr.get_num_free_gpu_blocks()
    mapping = mgr.swap_out(group)
    assert list(mapping.keys()) == gpu_blocks
    assert cpu_before == mgr.get_num_free_cpu_blocks() + len(gpu_blocks)
    assert gpu_before + len(gpu_blocks) == mgr.get_num_free_gpu_blocks()
    seq.status = SequenceStatus.SWAPPED
    cpu_blocks = mgr.get_block_table(seq)
    assert mgr.can_swap_in(group)
    cpu_before, gpu_before = mgr.get_num_free_cpu_blocks(), mgr.get_num_free_gpu_blocks()
    mapping = mgr.swap_in(group)
    assert list(mapping.keys()) == cpu_blocks
    assert cpu_before + len(cpu_blocks) == mgr.get_num_free_cpu_blocks()
    assert gpu_before == mgr.get_num_free_gpu_blocks() + len(cpu_blocks)


def test_free_returns_the_blocks_and_forgets_the_table():
    mgr = BlockSpaceManager(4, 4, 4, watermark=0)
    seq, group = _prompt("1", 4)
    mgr.allocate(group)
    held = len(mgr.get_block_table(seq))
    before = mgr.get_num_free_gpu_blocks()
    mgr.free(seq)
    assert mgr.get_num_free_gpu_blocks() == before + held
    with pytest.raises(KeyError):
        mgr.get_block_table(seq)


def test_reset_frees_everything():
    mgr = BlockSpaceManager(4, 4, 4, watermark=0)
    total = mgr.get_num_free_gpu_blocks()
    for i in range(4):
        _, group = _prompt(str(i), 4)
        mgr.allocate(group)
    assert mgr.get_num_free_gpu_blocks() == 0
    mgr.reset()
    assert mgr.get_num_free_gpu_blocks() == total


def test_block_numbers_are_fresh_first_then_first_freed_first():
    mgr = BlockSpaceManager(4, 6, 0, watermark=0)
    a, ga = _prompt("a", 8)
    b, gb = _prompt("b", 8)
    mgr.allocate(ga)
    mgr.allocate(gb)
    assert mgr.get_block_table(a) == [0, 1] and mgr.get_block_table(b) == [2, 3]
    mgr.free(a)                                                   # 0 and 1 go back, in table order
    c, gc = _prompt("c", 12)
    mgr.allocate(gc)
    assert mgr.get_block_table(c) == [4, 5, 0]                    # untouched ids before recycled ones
    d, gd = _prompt("d", 4)
    mgr.allocate(gd)
    assert mgr.get_block_table(d) == [1]
