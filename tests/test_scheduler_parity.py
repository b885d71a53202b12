"""Scheduler, block manager, slot mapper and input preparation against traces captured from the
REFERENCE's own Python (tools/gen_golden.py: vllm/core/scheduler.py with
track_prompt_blocks=True, vllm/worker/model_runner.py _prepare_prompt / _prepare_decode,
vllm/utils.py coalesce_blocks / SeqToSlotMapper).

Block numbers are compared exactly while the pool still hands out fresh blocks.  Once freed
blocks are reused the reference's order depends on Python object addresses (it frees a `set`
of block objects, vllm/core/block_manager.py:401-414), so from then on block-dependent fields
are checked for consistency (slot = table[i // bs] * bs + i % bs, runs cover the table) and
everything else exactly.
"""
import json
import os

import pytest

from helpers import GOLDEN
from splitwise_amd.config import CacheConfig, DeviceConfig, SchedulerConfig
from splitwise_amd.core.scheduler import Scheduler
from splitwise_amd.sampling_params import SamplingParams
from splitwise_amd.sequence import Logprob, Sequence, SequenceGroup, SequenceStatus
from splitwise_amd.utils import SeqToSlotMapper, coalesce_blocks
from splitwise_amd.worker.model_runner import ModelRunner


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def test_coalesce_blocks_matches_reference():
    for case in load("utils.json")["coalesce"]:
        assert [list(r) for r in coalesce_blocks(case["blocks"])] == case["runs"]
    assert coalesce_blocks([]) == []
    assert coalesce_blocks([5, 3, 4, 9, 10, 1]) == [(1, 1), (3, 3), (9, 2)]


def test_slot_mapper_matches_reference():
    mapper = SeqToSlotMapper()
    for op, seq_id, slot in load("utils.json")["slot_mapper"]:
        if op == "set":
            mapper.set_seq(seq_id)
            assert mapper.get_slot_id(seq_id) == slot
        else:
            mapper.free_seq(seq_id)


def _runs_cover(runs, table):
    blocks = sorted(b for start, n in runs for b in range(start, start + n))
    return blocks == sorted(table)


@pytest.mark.parametrize("name", ["basic", "preempt", "churn"])
def test_scheduler_trace_matches_reference(name):
    trace = load(f"scheduler_{name}.json")
    bs = trace["block_size"]
    sched_cfg = SchedulerConfig(trace["max_num_batched_tokens"], trace["max_num_seqs"],
                                trace["max_num_batched_tokens"], max_paddings=trace["max_paddings"])
    cache_cfg = CacheConfig(bs, 1.0, 1, "auto")
    cache_cfg.num_gpu_blocks = trace["num_gpu_blocks"]
    cache_cfg.num_cpu_blocks = trace["num_gpu_blocks"]
    sched = Scheduler(sched_cfg, cache_cfg, None, track_prompt_blocks=True, strict_reference=True)
    runner = ModelRunner(None, None, None, DeviceConfig("cpu"))
    runner.set_block_size(bs)
    lifetimes = {int(k): v for k, v in trace["lifetimes"].items()}
    seqs = {}
    t = 0.0
    blocks_exact = True          # until a freed block is handed out again
    for rec in trace["steps"]:
        for rid, plen in rec["arrivals"]:
            t += 1.0
            seq = Sequence(rid, "", list(range(1000 + rid, 1000 + rid + plen)), bs)
            group = SequenceGroup(str(rid), [seq], SamplingParams(temperature=0.0, max_tokens=10 ** 6), t)
            seqs[str(rid)] = (group, seq)
            sched.add_seq_group(group)
        pool = sched.block_manager.gpu_allocator
        metas, outs = sched.schedule()
        # once every block has been handed out, a freed block may have been reused in this step
        blocks_exact = blocks_exact and pool._next_fresh < pool.num_blocks
        assert bool(outs.prompt_run) == rec["prompt_run"], rec["step"]
        assert outs.num_batched_tokens == rec["num_batched_tokens"]
        assert [m.request_id for m in metas] == rec["scheduled"]
        assert [g.request_id for g in outs.ignored_seq_groups] == rec["ignored"]
        assert [g.request_id for g in sched.waiting] == rec["waiting"]
        assert [g.request_id for g in sched.running] == rec["running"]
        assert sched.block_manager.get_num_free_gpu_blocks() == rec["free_gpu_blocks"]
        tables = {m.request_id: list(next(iter(m.block_tables.values()))) for m in metas}
        assert {k: len(v) for k, v in tables.items()} == {k: len(v) for k, v in rec["block_tables"].items()}
        nw = {str(k): [list(r) for r in v] for k, v in outs.blocks_to_nw.items()}
        assert sorted(nw) == sorted(rec["blocks_to_nw"]), (rec["step"], nw, rec["blocks_to_nw"])
        assert {str(k): list(v) for k, v in outs.blocks_to_copy.items()} == rec["blocks_to_copy"] or not blocks_exact
        assert not outs.blocks_to_swap_in and not rec["blocks_to_swap_in"]
        assert not outs.blocks_to_swap_out and not rec["blocks_to_swap_out"]
        if blocks_exact:
            assert tables == rec["block_tables"], rec["step"]
            assert nw == rec["blocks_to_nw"], rec["step"]
        if metas:
            ref = rec["inputs"]
            if outs.prompt_run:
                toks, pos, md, plens, _ = runner._prepare_prompt(metas)
                assert toks.tolist() == ref["input_tokens"]
                assert pos.tolist() == ref["input_positions"]
                assert plens == ref["prompt_lens"] and md.max_seq_len == ref["max_seq_len"]
                slots = md.slot_mapping.tolist()
                if blocks_exact:
                    assert slots == ref["slot_mapping"]
                for row, m in zip(slots, metas):
                    table = tables[m.request_id]
                    n = len(next(iter(m.seq_data.values())).get_token_ids())
                    assert row[:n] == [table[i // bs] * bs + i % bs for i in range(n)]
                    assert all(s == -1 for s in row[n:])
            else:
                toks, pos, md = runner._prepare_decode(metas)
                assert toks.tolist() == ref["input_tokens"]
                assert pos.tolist() == ref["input_positions"]
                assert md.context_lens.tolist() == ref["context_lens"]
                assert md.max_context_len == ref["max_context_len"]
                if blocks_exact:
                    assert md.slot_mapping.tolist() == ref["slot_mapping"]
                    assert md.block_tables.tolist() == ref["block_tables"]
                for (slot,), (p,), m in zip(md.slot_mapping.tolist(), pos.tolist(), metas):
                    table = tables[m.request_id]
                    assert slot == table[p // bs] * bs + p % bs
        # the slot a sequence's blocks are announced under covers exactly its block table
        slot_of = sched.seq_to_slot_mapper
        for m in metas:
            seq_id = next(iter(m.seq_data))
            data = m.seq_data[seq_id]
            if data.get_output_len() <= 1:
                assert _runs_cover(outs.blocks_to_nw[slot_of.get_slot_id(seq_id)], tables[m.request_id])
        finished = []
        for m in metas:
            group, seq = seqs[m.request_id]
            seq.append_token_id(7, {7: Logprob(0.0)})
            if seq.get_output_len() >= lifetimes[int(m.request_id)]:
                seq.status = SequenceStatus.FINISHED_STOPPED
                sched.free_seq(seq)
                finished.append(m.request_id)
        sched.free_finished_seq_groups()
        assert finished == rec["finished"]
        if pool._next_fresh >= pool.num_blocks:
            blocks_exact = False


def test_prefix_caching_trace_matches_reference():
    """Scheduler + BlockSpaceManager(enable_caching=True) + _prepare_prompt over cached prefixes, step by step
    against the reference's own run (tools/gen_golden.py gen_prefix_caching): shared prompt blocks, computed
    block lists, blocks revived from the evictor, least-recently-used eviction once the pool is full.
    Everything that does not name a physical block is compared exactly throughout.  Block numbers are compared
    exactly until the first EVICTION: from there the reference's choice among blocks of equal age and
    equal hashed-token count depends on the iteration order of a set of Python objects
    (vllm/core/block_manager.py:401-414), so tables are checked for shape and consistency instead."""
    trace = load("scheduler_prefix_caching.json")
    bs = trace["block_size"]
    sched_cfg = SchedulerConfig(trace["max_num_batched_tokens"], trace["max_num_seqs"],
                                trace["max_num_batched_tokens"], max_paddings=trace["max_paddings"])
    cache_cfg = CacheConfig(bs, 1.0, 1, "auto", enable_prefix_caching=True)
    cache_cfg.num_gpu_blocks = trace["num_gpu_blocks"]
    cache_cfg.num_cpu_blocks = trace["num_gpu_blocks"]
    sched = Scheduler(sched_cfg, cache_cfg, None, track_prompt_blocks=True, strict_reference=True)
    runner = ModelRunner(None, None, None, DeviceConfig("cpu"))
    runner.set_block_size(bs)
    lifetimes = {int(k): v for k, v in trace["lifetimes"].items()}
    pool = sched.block_manager.gpu_allocator
    seqs = {}
    t = 0.0
    exact_steps = 0
    evicted_yet = False
    for rec in trace["steps"]:
        for rid, tokens in rec["arrivals"]:
            t += 1.0
            seq = Sequence(rid, "", list(tokens), bs)
            group = SequenceGroup(str(rid), [seq], SamplingParams(temperature=0.0, max_tokens=10 ** 6), t)
            seqs[str(rid)] = (group, seq)
            sched.add_seq_group(group)
        fresh_before = pool.current_num_blocks
        free_before = len(pool.free_table)
        metas, outs = sched.schedule()
        # an eviction = a block handed out when no fresh one was left
        evicted = fresh_before == pool.num_blocks and len(pool.free_table) < free_before
        evicted_yet = evicted_yet or evicted
        blocks_exact = not evicted_yet
        step = rec["step"]
        assert bool(outs.prompt_run) == rec["prompt_run"], step
        assert [m.request_id for m in metas] == rec["scheduled"], step
        assert [g.request_id for g in sched.waiting] == rec["waiting"], step
        assert [g.request_id for g in sched.running] == rec["running"], step
        assert sched.block_manager.get_num_free_gpu_blocks() == rec["free_gpu_blocks"], step
        tables = {m.request_id: list(next(iter(m.block_tables.values()))) for m in metas}
        computed = {m.request_id: list(m.computed_block_nums or []) for m in metas}
        assert {k: len(v) for k, v in tables.items()} == {k: len(v) for k, v in rec["block_tables"].items()}, step
        assert {k: len(v) for k, v in computed.items()} == {k: len(v) for k, v in rec["computed_block_nums"].items()}, step
        for rid, ids in computed.items():
            assert ids == tables[rid][:len(ids)]                   # a leading run of the table, never all of it
            assert len(ids) < len(tables[rid])
        nw = {str(k): [list(r) for r in v] for k, v in outs.blocks_to_nw.items()}
        assert sorted(nw) == sorted(rec["blocks_to_nw"]), step
        if blocks_exact:
            exact_steps += 1
            assert tables == rec["block_tables"], step
            assert computed == rec["computed_block_nums"], step
            assert nw == rec["blocks_to_nw"], step
            assert {str(k): list(v) for k, v in outs.blocks_to_copy.items()} == rec["blocks_to_copy"], step
        if metas and outs.prompt_run:
            ref = rec["inputs"]
            toks, pos, md, plens, sublens = runner._prepare_prompt(metas)
            assert toks.tolist() == ref["input_tokens"], step
            assert pos.tolist() == ref["input_positions"], step
            assert plens == ref["prompt_lens"] and sublens == ref["subquery_lens"], step
            assert md.max_seq_len == ref["max_seq_len"] and md.start_loc.tolist() == ref["start_loc"], step
            assert md.context_lens.tolist() == ref["context_lens"], step
            assert list(md.block_tables.shape) == [len(ref["block_tables"]), len(ref["block_tables"][0])], step
            if blocks_exact:
                assert md.slot_mapping.tolist() == ref["slot_mapping"], step
                assert md.block_tables.tolist() == ref["block_tables"], step
            for row, m, ctx in zip(md.slot_mapping.tolist(), metas, md.context_lens.tolist()):
                table = tables[m.request_id]
                n = len(next(iter(m.seq_data.values())).get_token_ids())
                assert row[:n - ctx] == [table[i // bs] * bs + i % bs for i in range(ctx, n)]
        for m in metas:
            sched.mark_blocks_as_computed(seqs[m.request_id][0])
        finished = []
        for m in metas:
            group, seq = seqs[m.request_id]
            tok = 7 + int(m.request_id) % 2
            seq.append_token_id(tok, {tok: Logprob(0.0)})
            if seq.get_output_len() >= lifetimes[int(m.request_id)]:
                seq.status = SequenceStatus.FINISHED_STOPPED
                sched.free_seq(seq)
                finished.append(m.request_id)
        sched.free_finished_seq_groups()
        assert finished == rec["finished"], step
    assert exact_steps >= 16        # sharing, promotion onto an existing block and revival are all inside the exact part
    assert sched.block_manager.get_num_free_gpu_blocks() == trace["num_gpu_blocks"]
