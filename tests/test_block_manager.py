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


# ---------------------------------------------------------------- prefix caching
def test_caching_pool_shares_blocks_by_hash_and_revives_freed_ones():
    """Scenario and answers of the reference's tests/prefix_caching/test_prefix_caching.py:13-43."""
    from splitwise_amd.core.block_manager import CachingBlockPool
    from splitwise_amd.utils import Device
    pool = CachingBlockPool(Device.CPU, 16, 16)
    first = pool.allocate(1, 0)
    second = pool.allocate(1, 0)
    assert first == second and pool.ref_count(second) == 2
    pool.free(first)
    assert pool.ref_count(second) == 1
    pool.free(second)
    again = pool.allocate(1, 0)                      # its count went to 0, yet the same block comes back
    assert again == second and pool.block_hash[again] == 1


def test_caching_pool_evicts_the_least_recently_used_block():
    """tests/prefix_caching/test_prefix_caching.py:46-85."""
    from splitwise_amd.core.block_manager import CachingBlockPool
    from splitwise_amd.utils import Device
    num_blocks, block_size = 16, 16
    pool = CachingBlockPool(Device.CPU, block_size, num_blocks)
    blocks = [pool.allocate(i, 0) for i in range(num_blocks)]
    for block in blocks:
        pool.free(block)
    new = pool.allocate(block_size, 0)
    assert new == blocks[0] and pool.block_hash[new] == block_size       # the first one freed
    realloc = pool.allocate(1, 0)                                         # taken back out of the free list
    assert realloc == blocks[1] and pool.block_hash[realloc] == 1
    new = pool.allocate(block_size + 1, 0)
    assert new != realloc and pool.block_hash[new] == block_size + 1 and new == 2


def test_eviction_prefers_older_blocks_then_the_one_that_hashes_more_tokens():
    """vllm/core/evictor.py:68-104."""
    from splitwise_amd.core.block_manager import CachingBlockPool
    from splitwise_amd.utils import Device
    pool = CachingBlockPool(Device.GPU, 4, 3)
    a, b, c = pool.allocate(10, 4), pool.allocate(11, 8), pool.allocate(12, 12)
    pool.last_accessed[a], pool.last_accessed[b], pool.last_accessed[c] = 5.0, 3.0, 3.0
    pool.computed[c] = True
    for block in (a, b, c):
        pool.free(block)
    assert pool.allocate(20, 4) == c and not pool.computed[c]            # oldest stamp, most hashed tokens
    assert pool.allocate(21, 4) == b
    assert pool.allocate(22, 4) == a
    with pytest.raises(ValueError, match="No usable cache memory left"):
        pool.allocate(23, 4)


def test_prompts_with_a_common_prefix_share_blocks_and_report_them_computed():
    """block_manager.py:199-201 (allocation by content hash), :441-487 (computed blocks): the second prompt
    gets the first one's full blocks, and may skip those that have been marked computed -- which, as the
    reference writes it (:441-451), are the full blocks of a sequence except the last full one."""
    bm = BlockSpaceManager(4, 16, 16, watermark=0, enable_caching=True)
    prefix = list(range(100, 114))                                        # 3 full blocks + 2 tokens
    a = Sequence(1, None, prefix + [1], 4)                                # the fourth block holds 3 of 4 tokens
    ga = SequenceGroup("a", [a], SamplingParams(), time.time())
    bm.allocate(ga)
    a.status = SequenceStatus.RUNNING
    assert bm.get_common_computed_block_ids(ga) == []                     # nothing has been computed yet
    bm.mark_blocks_as_computed(ga)                                        # the prompt step has run
    b = Sequence(2, None, prefix + [7, 8, 9], 4)
    gb = SequenceGroup("b", [b], SamplingParams(), time.time())
    bm.allocate(gb)
    ta, tb = bm.get_block_table(a), bm.get_block_table(b)
    assert ta[:3] == tb[:3] and ta[3] != tb[3]                            # the block where they diverge is private
    assert bm.get_common_computed_block_ids(gb) == tb[:2]
    assert bm.gpu_allocator.ref_count(ta[0]) == 2
    # an identical prompt shares every block, the partly filled last one too
    c = Sequence(3, None, prefix + [1], 4)
    gc = SequenceGroup("c", [c], SamplingParams(), time.time())
    bm.allocate(gc)
    assert bm.get_block_table(c) == ta and bm.gpu_allocator.ref_count(ta[3]) == 2
    assert bm.get_common_computed_block_ids(gc) == ta[:2]
    # the first append to the shared last block copies it
    c.status = SequenceStatus.RUNNING
    _append(c, 5)
    src_dst = bm.append_slot(c)
    assert src_dst is not None and src_dst[0] == ta[3] and bm.get_block_table(c)[3] == src_dst[1]
    for seq in (a, b, c):
        bm.free(seq)
    assert bm.get_num_free_gpu_blocks() == 16
    # freed blocks stay findable: the same prefix comes back on the same blocks, still computed
    d = Sequence(4, None, prefix + [3], 4)
    gd = SequenceGroup("d", [d], SamplingParams(), time.time())
    bm.allocate(gd)
    assert bm.get_block_table(d)[:3] == ta[:3] and bm.get_common_computed_block_ids(gd) == ta[:2]
    # a prompt that is cached completely still runs its last block
    e = Sequence(5, None, prefix[:8], 4)
    ge = SequenceGroup("e", [e], SamplingParams(), time.time())
    bm.allocate(ge)
    assert bm.get_block_table(e) == ta[:2] and bm.get_common_computed_block_ids(ge) == ta[:1]


def test_a_block_that_fills_up_is_promoted_to_its_content_hash():
    """block_manager.py:218-252: two sequences that grow into the same content end up on one block."""
    bm = BlockSpaceManager(4, 16, 16, watermark=0, enable_caching=True)
    seqs = []
    for i in (1, 2):
        s = Sequence(i, None, [9, 9, 9, 9, 5] if i == 1 else [9, 9, 9, 9, 5, 6], 4)
        g = SequenceGroup(str(i), [s], SamplingParams(), time.time())
        bm.allocate(g)
        s.status = SequenceStatus.RUNNING
        seqs.append(s)
    s1, s2 = seqs
    assert bm.get_block_table(s1)[0] == bm.get_block_table(s2)[0]
    assert bm.get_block_table(s1)[1] != bm.get_block_table(s2)[1]         # different partial contents
    for tok in (6, 7, 8):
        _append(s1, tok)
        assert bm.append_slot(s1) is None
    for tok in (7, 8):
        _append(s2, tok)
        assert bm.append_slot(s2) is None
    # both second blocks now hold 5 6 7 8 after the same first block: one physical block
    assert bm.get_block_table(s1)[1] == bm.get_block_table(s2)[1]
    assert bm.gpu_allocator.ref_count(bm.get_block_table(s1)[1]) == 2


@pytest.mark.parametrize("seed", range(6))
def test_placeholders_never_make_two_sequences_share_a_block_they_fill_differently(seed):
    """Pipelined steps: room is made for a token before it is known (the sequence ends in a placeholder, -1), and
    the token before it is filled in afterwards.  Random sequences over a few common prompts, each drawing its own
    tokens: whenever two live sequences hold the same physical block at some index, the tokens they know of in it
    must agree; reference counts equal the number of tables that hold a block; a block that filled up under a
    placeholder is found by its content once the token is known; and everything comes back."""
    import random

    from splitwise_amd.sequence import PLACEHOLDER_TOKEN_ID
    rng = random.Random(seed)
    bs, blocks = 4, 64
    bm = BlockSpaceManager(bs, blocks, 8, watermark=0, enable_caching=True)
    heads = [[rng.randrange(3) for _ in range(n)] for n in (3, 4, 7, 8, 9)]
    live, next_id = [], [1000 * (seed + 1)]

    def check():
        holders = {}
        for seq in live:
            table = bm.get_block_table(seq)
            assert len(table) == seq.num_logical_blocks
            ids = seq.get_token_ids()
            for k, block in enumerate(dict.fromkeys(table)):
                known = tuple(t for t in ids[bs * k:bs * (k + 1)] if t != PLACEHOLDER_TOKEN_ID)
                holders.setdefault(block, []).append((k, known))
        for block, held in holders.items():
            assert bm.gpu_allocator.ref_count(block) == len(held), (block, held)
            for k, known in held[1:]:
                k0, known0 = held[0]
                n = min(len(known), len(known0))
                assert k == k0 and known[:n] == known0[:n], (block, held)

    for _ in range(400):
        roll = rng.random()
        if (roll < 0.15 or not live) and len(live) < 8:
            next_id[0] += 1
            seq = Sequence(next_id[0], None, list(rng.choice(heads)), bs)
            bm.allocate(SequenceGroup(str(next_id[0]), [seq], SamplingParams(), time.time()))
            seq.status = SequenceStatus.RUNNING
            live.append(seq)
        elif roll < 0.22:
            bm.free(live.pop(rng.randrange(len(live))))
        else:
            seq = rng.choice(live)
            if seq.get_len() >= 40:
                continue
            # one pipelined step: the next token's place is made first, the previous token arrives afterwards
            had = seq.has_placeholder()
            if had and rng.random() < 0.5:
                tok = rng.randrange(3)
                seq.fill_placeholder(tok, {tok: Logprob(0.0)})
                had = False
            if not had:
                seq.append_placeholder()
                bm.append_slot(seq)
        check()
    # a finished sequence's full blocks can be found again by a prompt with the same tokens
    for seq in live:
        if seq.has_placeholder():
            tok = rng.randrange(3)
            seq.fill_placeholder(tok, {tok: Logprob(0.0)})
    donor = max(live, key=lambda s: s.get_len())
    if donor.get_len() >= 3 * bs:
        donor.append_placeholder()
        bm.append_slot(donor)                                  # (hashes the block that filled up one step ago)
        ids = [t for t in donor.get_token_ids() if t != PLACEHOLDER_TOKEN_ID]
        twin = Sequence(next_id[0] + 1, None, ids[:(len(ids) // bs) * bs] + [7], bs)
        bm.allocate(SequenceGroup("twin", [twin], SamplingParams(), time.time()))
        full = len(ids) // bs
        # (the donor's own block, or an older one with the same content: blocks are hashed in place, never merged)
        shared = sum(a == b for a, b in zip(bm.get_block_table(twin)[:full], bm.get_block_table(donor)[:full]))
        assert shared >= 1 and all(bm.gpu_allocator.block_hash[b] == donor.hash_of_block(k)
                                   for k, b in enumerate(bm.get_block_table(twin)[:full]))
        live.append(twin)
        check()
    for seq in live:
        bm.free(seq)
    assert bm.get_num_free_gpu_blocks() == blocks and not bm._postponed
