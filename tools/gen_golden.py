#!/usr/bin/env python3
"""Generates tests/golden/* by running the REFERENCE's own Python on seeded inputs.

Runs only in the build container, where the reference checkout is mounted at
/root/reference.  Absent third-party / compiled modules (GPUtil, vllm._C, xformers)
are replaced by empty stand-ins so the pure-Python parts import; every value
written to a fixture is computed by the reference's code:

  paged_attention_*.npz   tests/kernels/test_attention.py  ref_single_query_cached_kv_attention
  prefill_attention_*.npz tests/kernels/test_attention.py  ref_multi_query_kv_attention
  scheduler_*.json        vllm/core/scheduler.py Scheduler(track_prompt_blocks=True) step traces
                          + vllm/worker/model_runner.py _prepare_prompt / _prepare_decode tensors
  utils.json              vllm/utils.py coalesce_blocks, SeqToSlotMapper

Usage:  python tools/gen_golden.py          (writes into tests/golden/)
"""
import json
import os
import random
import sys
import types

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "tests", "kernels"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_stub("GPUtil", getGPUs=lambda: [])
_ns = types.SimpleNamespace
_stub("vllm._C", ops=_ns(), cache_ops=_ns(),
      cuda_utils=_ns(get_max_shared_memory_per_block_device_attribute=lambda g: 65536))
_x = _stub("xformers")
_xo = _stub("xformers.ops")
_x.ops = _xo
_f = _stub("xformers.ops.fmha")
_xo.fmha = _f
_f.attn_bias = _stub("xformers.ops.fmha.attn_bias", BlockDiagonalCausalMask=object,
                     LowerTriangularMaskWithTensorBias=object)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_attention as ref_test_attention  # noqa: E402  (reference test module)
from vllm.config import CacheConfig, SchedulerConfig, DeviceConfig  # noqa: E402
from vllm.core.scheduler import Scheduler  # noqa: E402
from vllm.sampling_params import SamplingParams  # noqa: E402
from vllm.sequence import Logprob, Sequence, SequenceGroup, SequenceStatus  # noqa: E402
from vllm.utils import SeqToSlotMapper, coalesce_blocks  # noqa: E402
from vllm.worker.model_runner import ModelRunner  # noqa: E402

from oracle import cache_ops as oracle_cache  # noqa: E402  (layout writer; gather is the reference's)

_NP = {torch.float16: np.float16, torch.float32: np.float32}


def _to_np(t: torch.Tensor) -> np.ndarray:
    if t.dtype == torch.bfloat16:          # npz has no bf16: store the raw bits
        return t.contiguous().view(torch.int16).numpy().view(np.uint16)
    return t.contiguous().numpy()


# ------------------------------------------------------------------ attention
def gen_paged_attention():
    cases = [
        # name, dtype, (H, KVH), head, block, alibi, num_seqs, max_ctx, num_blocks
        ("f16_mha_d128_b16", torch.float16, (4, 4), 128, 16, False, 5, 300, 24),
        ("bf16_gqa8_d128_b16", torch.bfloat16, (16, 2), 128, 16, False, 4, 700, 48),
        ("bf16_mha_d64_b32_alibi", torch.bfloat16, (4, 4), 64, 32, True, 3, 260, 12),
        ("f16_gqa4_d80_b8_alibi", torch.float16, (8, 2), 80, 8, True, 3, 130, 20),
        ("f32_mha_d96_b16", torch.float32, (4, 4), 96, 16, False, 3, 200, 16),
        ("f16_mha_d256_b16_long", torch.float16, (1, 1), 256, 16, False, 2, 1300, 84),
        ("bf16_gqa2_d112_b32", torch.bfloat16, (4, 2), 112, 32, False, 3, 150, 8),
    ]
    for name, dtype, (h, kvh), head, bs, alibi, num_seqs, max_ctx, num_blocks in cases:
        random.seed(0)
        torch.manual_seed(0)
        scale = float(1.0 / head ** 0.5)
        query = torch.empty(num_seqs, h, head, dtype=dtype).uniform_(-scale, scale)
        slopes = torch.randn(h, dtype=torch.float) if alibi else None
        ctx = [random.randint(1, max_ctx) for _ in range(num_seqs)]
        ctx[-1] = max_ctx
        max_blocks = (max_ctx + bs - 1) // bs
        tables = [[random.randint(0, num_blocks - 1) for _ in range(max_blocks)] for _ in range(num_seqs)]
        x = 16 // torch.tensor([], dtype=dtype).element_size()
        key_cache = torch.empty(num_blocks, kvh, head // x, bs, x, dtype=dtype).uniform_(-scale, scale)
        value_cache = torch.empty(num_blocks, kvh, head, bs, dtype=dtype).uniform_(-scale, scale)
        out = torch.empty_like(query)
        ref_test_attention.ref_single_query_cached_kv_attention(
            out, query, h // kvh, key_cache, value_cache, torch.tensor(tables, dtype=torch.int),
            torch.tensor(ctx, dtype=torch.int), scale, slopes)
        np.savez_compressed(
            os.path.join(OUT, f"paged_attention_{name}.npz"),
            dtype=str(dtype).replace("torch.", ""), num_kv_heads=kvh, scale=scale, block_size=bs,
            query=_to_np(query), key_cache=_to_np(key_cache), value_cache=_to_np(value_cache),
            block_tables=np.array(tables, dtype=np.int32), context_lens=np.array(ctx, dtype=np.int32),
            alibi_slopes=(slopes.numpy() if alibi else np.zeros(0, np.float32)), out=_to_np(out))
        print("paged_attention", name)


def gen_prefill_attention():
    cases = [
        ("f16_mha_d128", torch.float16, (4, 4), 128, [37, 128, 5]),
        ("bf16_gqa4_d128", torch.bfloat16, (8, 2), 128, [200, 1, 64]),
        ("f16_mha_d64", torch.float16, (3, 3), 64, [77, 130]),
        ("bf16_gqa8_d80", torch.bfloat16, (8, 1), 80, [90, 33]),
    ]
    for name, dtype, (h, kvh), head, seq_lens in cases:
        torch.manual_seed(0)
        scale = float(1.0 / head ** 0.5)
        n = sum(seq_lens)
        qkv = torch.empty(n, h + 2 * kvh, head, dtype=dtype).uniform_(-scale, scale)
        q, k, v = qkv.split([h, kvh, kvh], dim=1)
        g = h // kvh
        kk = torch.repeat_interleave(k, g, dim=1) if g > 1 else k
        vv = torch.repeat_interleave(v, g, dim=1) if g > 1 else v
        cu = [0]
        for s in seq_lens:
            cu.append(cu[-1] + s)
        out = ref_test_attention.ref_multi_query_kv_attention(cu, q, kk, vv, scale, dtype)
        np.savez_compressed(
            os.path.join(OUT, f"prefill_attention_{name}.npz"),
            dtype=str(dtype).replace("torch.", ""), scale=scale, seq_lens=np.array(seq_lens),
            query=_to_np(q), key=_to_np(k), value=_to_np(v), out=_to_np(out))
        print("prefill_attention", name)


def gen_cache_layout():
    """K/V rows written into the paged layout, then read back through the
    reference's own gather (tests/kernels/test_attention.py:84-93): pins the
    [block, head, d/x, offset, x] / [block, head, d, offset] index rule."""
    random.seed(1)
    torch.manual_seed(1)
    dtype, h, head, bs, nb = torch.float16, 4, 64, 16, 32
    x = 8
    scale = head ** -0.5
    ctx = 45
    table = random.sample(range(nb), (ctx + bs - 1) // bs)
    key = torch.empty(ctx, h, head, dtype=dtype).uniform_(-1, 1)
    value = torch.empty(ctx, h, head, dtype=dtype).uniform_(-1, 1)
    slots = torch.tensor([table[i // bs] * bs + i % bs for i in range(ctx)], dtype=torch.long)
    kc = torch.zeros(nb, h, head // x, bs, x, dtype=dtype)
    vc = torch.zeros(nb, h, head, bs, dtype=dtype)
    oracle_cache.reshape_and_cache(key, value, kc, vc, slots)
    query = torch.empty(1, h, head, dtype=dtype).uniform_(-1, 1)
    out = torch.empty_like(query)
    ref_test_attention.ref_single_query_cached_kv_attention(
        out, query, 1, kc, vc, torch.tensor([table], dtype=torch.int), torch.tensor([ctx], dtype=torch.int),
        scale, None)
    # the same attention straight from the un-paged rows (reference's ref_masked_attention)
    direct = ref_test_attention.ref_masked_attention(query, key, value, scale)
    assert torch.equal(out, direct.view_as(out)), "paged gather disagrees with the contiguous rows"
    np.savez_compressed(os.path.join(OUT, "cache_layout_f16.npz"), key=_to_np(key), value=_to_np(value),
                        slots=slots.numpy(), table=np.array(table, dtype=np.int32), query=_to_np(query),
                        out=_to_np(out), block_size=bs, num_blocks=nb, scale=scale)
    print("cache_layout")


# ------------------------------------------------------------------ scheduler
def _mk_group(request_id, prompt_len, block_size, arrival):
    seq = Sequence(int(request_id), "", list(range(1000 + int(request_id), 1000 + int(request_id) + prompt_len)),
                   block_size)
    return SequenceGroup(str(request_id), [seq], SamplingParams(temperature=0.0, max_tokens=10 ** 6),
                         arrival), seq


def _trace_scenario(name, block_size, num_gpu_blocks, max_num_seqs, max_batched, arrivals, lifetimes,
                    max_steps=200):
    """arrivals: {step: [(request_id, prompt_len)]}; lifetimes: {request_id: output tokens}."""
    sched_cfg = SchedulerConfig(max_batched, max_num_seqs, max_batched, max_paddings=256)
    cache_cfg = CacheConfig(block_size, 1.0, 1, "auto")
    cache_cfg.num_gpu_blocks = num_gpu_blocks
    cache_cfg.num_cpu_blocks = num_gpu_blocks
    sched = Scheduler(sched_cfg, cache_cfg, None, track_prompt_blocks=True)
    runner = ModelRunner(None, None, None, DeviceConfig("cpu"), None)
    runner.set_block_size(block_size)
    runner.model_config = types.SimpleNamespace(enforce_eager=True)
    seqs = {}
    steps = []
    t = 0.0
    for step in range(max_steps):
        for rid, plen in arrivals.get(step, []):
            t += 1.0
            group, seq = _mk_group(rid, plen, block_size, t)
            seqs[str(rid)] = (group, seq)
            sched.add_seq_group(group)
        if not sched.has_unfinished_seqs() and step > max(arrivals):
            break
        metas, outs = sched.schedule()
        rec = {
            "step": step,
            "arrivals": [[int(r), int(p)] for r, p in arrivals.get(step, [])],
            "prompt_run": bool(outs.prompt_run),
            "num_batched_tokens": int(outs.num_batched_tokens),
            "scheduled": [m.request_id for m in metas],
            "block_tables": {m.request_id: list(next(iter(m.block_tables.values()))) for m in metas},
            "blocks_to_nw": {str(k): [list(r) for r in v] for k, v in outs.blocks_to_nw.items()},
            "blocks_to_copy": {str(k): list(v) for k, v in outs.blocks_to_copy.items()},
            "blocks_to_swap_in": {str(k): int(v) for k, v in outs.blocks_to_swap_in.items()},
            "blocks_to_swap_out": {str(k): int(v) for k, v in outs.blocks_to_swap_out.items()},
            "ignored": [g.request_id for g in outs.ignored_seq_groups],
            "waiting": [g.request_id for g in sched.waiting],
            "running": [g.request_id for g in sched.running],
            "free_gpu_blocks": sched.block_manager.get_num_free_gpu_blocks(),
        }
        if metas:
            if outs.prompt_run:
                toks, pos, meta, plens, sublens, *_ = runner._prepare_prompt(metas)
                rec["inputs"] = {
                    "input_tokens": toks.tolist(), "input_positions": pos.tolist(),
                    "slot_mapping": meta.slot_mapping.tolist(), "prompt_lens": list(plens),
                    "max_seq_len": int(meta.max_seq_len),
                }
            else:
                toks, pos, meta, *_ = runner._prepare_decode(metas)
                rec["inputs"] = {
                    "input_tokens": toks.tolist(), "input_positions": pos.tolist(),
                    "slot_mapping": meta.slot_mapping.tolist(), "context_lens": meta.context_lens.tolist(),
                    "block_tables": meta.block_tables.tolist(), "max_context_len": int(meta.max_context_len),
                }
        # every scheduled sequence emits one token; finished ones are freed (LLMEngine does the same)
        finished = []
        for m in metas:
            group, seq = seqs[m.request_id]
            seq.append_token_id(7, {7: Logprob(0.0)})
            if seq.get_output_len() >= lifetimes[int(m.request_id)]:
                seq.status = SequenceStatus.FINISHED_STOPPED
                sched.free_seq(seq)
                finished.append(m.request_id)
        sched.free_finished_seq_groups()
        rec["finished"] = finished
        steps.append(rec)
    with open(os.path.join(OUT, f"scheduler_{name}.json"), "w") as f:
        json.dump({"block_size": block_size, "num_gpu_blocks": num_gpu_blocks, "max_num_seqs": max_num_seqs,
                   "max_num_batched_tokens": max_batched, "max_paddings": 256,
                   "lifetimes": {str(k): v for k, v in lifetimes.items()}, "steps": steps}, f)
    print("scheduler", name, len(steps), "steps")


def gen_scheduler():
    # three prompts at once, then a late arrival while the others decode
    _trace_scenario("basic", 16, 64, 8, 256,
                    {0: [(0, 10), (1, 20), (2, 40)], 3: [(3, 33)]},
                    {0: 4, 1: 7, 2: 3, 3: 5})
    # block pressure: preemption by recompute, re-prefill, slot reuse
    _trace_scenario("preempt", 4, 14, 8, 64,
                    {0: [(0, 9), (1, 11), (2, 7)], 2: [(3, 6)], 9: [(4, 5)]},
                    {0: 12, 1: 9, 2: 14, 3: 6, 4: 3})
    # many short requests churning slots
    rng = random.Random(3)
    arrivals, lifetimes = {}, {}
    rid = 0
    for step in range(0, 30, 2):
        arrivals[step] = []
        for _ in range(rng.randint(1, 3)):
            arrivals[step].append((rid, rng.randint(1, 70)))
            lifetimes[rid] = rng.randint(1, 9)
            rid += 1
    _trace_scenario("churn", 16, 40, 6, 128, arrivals, lifetimes)


def _trace_prefix_scenario(name, block_size, num_gpu_blocks, max_num_seqs, max_batched, arrivals, lifetimes,
                           max_steps=200):
    """Scheduler + BlockSpaceManager(enable_caching=True) + _prepare_prompt with cached prefixes.
    arrivals: {step: [(request_id, prompt token ids)]}.  After every step the scheduled groups are marked
    computed, as LLMEngine._process_model_outputs does (vllm/engine/llm_engine.py:543-547)."""
    sched_cfg = SchedulerConfig(max_batched, max_num_seqs, max_batched, max_paddings=256)
    cache_cfg = CacheConfig(block_size, 1.0, 1, "auto", enable_prefix_caching=True)
    cache_cfg.num_gpu_blocks = num_gpu_blocks
    cache_cfg.num_cpu_blocks = num_gpu_blocks
    sched = Scheduler(sched_cfg, cache_cfg, None, track_prompt_blocks=True)
    runner = ModelRunner(None, None, None, DeviceConfig("cpu"), None)
    runner.set_block_size(block_size)
    runner.model_config = types.SimpleNamespace(enforce_eager=True)
    seqs = {}
    steps = []
    t = 0.0
    for step in range(max_steps):
        for rid, tokens in arrivals.get(step, []):
            t += 1.0
            seq = Sequence(int(rid), "", list(tokens), block_size)
            group = SequenceGroup(str(rid), [seq], SamplingParams(temperature=0.0, max_tokens=10 ** 6), t)
            seqs[str(rid)] = (group, seq)
            sched.add_seq_group(group)
        if not sched.has_unfinished_seqs() and step > max(arrivals):
            break
        metas, outs = sched.schedule()
        rec = {
            "step": step,
            "arrivals": [[int(r), list(map(int, toks))] for r, toks in arrivals.get(step, [])],
            "prompt_run": bool(outs.prompt_run),
            "scheduled": [m.request_id for m in metas],
            "block_tables": {m.request_id: list(next(iter(m.block_tables.values()))) for m in metas},
            "computed_block_nums": {m.request_id: list(m.computed_block_nums or []) for m in metas},
            "blocks_to_nw": {str(k): [list(r) for r in v] for k, v in outs.blocks_to_nw.items()},
            "blocks_to_copy": {str(k): list(v) for k, v in outs.blocks_to_copy.items()},
            "waiting": [g.request_id for g in sched.waiting],
            "running": [g.request_id for g in sched.running],
            "free_gpu_blocks": sched.block_manager.get_num_free_gpu_blocks(),
        }
        if metas and outs.prompt_run:
            toks, pos, meta, plens, sublens, *_ = runner._prepare_prompt(metas)
            rec["inputs"] = {
                "input_tokens": toks.tolist(), "input_positions": pos.tolist(),
                "slot_mapping": meta.slot_mapping.tolist(), "prompt_lens": list(plens),
                "subquery_lens": list(sublens), "max_seq_len": int(meta.max_seq_len),
                "context_lens": meta.context_lens.tolist(), "block_tables": meta.block_tables.tolist(),
                "start_loc": meta.start_loc.tolist(),
            }
        for m in metas:
            sched.mark_blocks_as_computed(seqs[m.request_id][0])
        finished = []
        for m in metas:
            group, seq = seqs[m.request_id]
            tok = 7 + int(m.request_id) % 2           # two output streams: some continuations coincide
            seq.append_token_id(tok, {tok: Logprob(0.0)})
            if seq.get_output_len() >= lifetimes[int(m.request_id)]:
                seq.status = SequenceStatus.FINISHED_STOPPED
                sched.free_seq(seq)
                finished.append(m.request_id)
        sched.free_finished_seq_groups()
        rec["finished"] = finished
        steps.append(rec)
    with open(os.path.join(OUT, f"scheduler_{name}.json"), "w") as f:
        json.dump({"block_size": block_size, "num_gpu_blocks": num_gpu_blocks, "max_num_seqs": max_num_seqs,
                   "max_num_batched_tokens": max_batched, "max_paddings": 256, "enable_prefix_caching": True,
                   "lifetimes": {str(k): v for k, v in lifetimes.items()}, "steps": steps}, f)
    print("scheduler", name, len(steps), "steps")


def gen_prefix_caching():
    prefix = list(range(500, 513))                      # 13 tokens: three full blocks of 4 and one token
    other = list(range(900, 931))
    arrivals = {
        0: [(0, prefix + [1, 2, 3])],
        2: [(1, prefix + [4, 5, 6, 7, 8])],             # shares the three computed blocks
        3: [(2, prefix + [1, 2, 3])],                   # the same prompt as request 0, which still runs
        6: [(3, other)],                                # unrelated
        12: [(4, prefix + [9, 9])],                     # the sharers have finished: blocks come back from the evictor
        16: [(5, list(range(2000, 2041)))],             # fills the pool: least recently used blocks are evicted
        17: [(6, list(range(3000, 3030)))],
        30: [(7, prefix + [1, 2, 3, 4])],               # after the eviction
        31: [(8, other + [5])],
    }
    lifetimes = {0: 7, 1: 4, 2: 6, 3: 9, 4: 3, 5: 8, 6: 5, 7: 4, 8: 3}
    _trace_prefix_scenario("prefix_caching", 4, 26, 8, 128, arrivals, lifetimes)


def gen_utils():
    rng = random.Random(5)
    cases = []
    for _ in range(40):
        n = rng.randint(0, 30)
        blocks = rng.sample(range(60), n)
        cases.append({"blocks": blocks, "runs": [list(r) for r in coalesce_blocks(list(blocks))]})
    mapper = SeqToSlotMapper()
    ops, live = [], []
    for i in range(200):
        if live and rng.random() < 0.45:
            sid = live.pop(rng.randrange(len(live)))
            mapper.free_seq(sid)
            ops.append(["free", sid, -1])
        else:
            mapper.set_seq(i)
            live.append(i)
            ops.append(["set", i, mapper.get_slot_id(i)])
    with open(os.path.join(OUT, "utils.json"), "w") as f:
        json.dump({"coalesce": cases, "slot_mapper": ops}, f)
    print("utils")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])                 # e.g. `gen_golden.py prefix_caching`; nothing = everything
    for tag, fn in (("paged_attention", gen_paged_attention), ("prefill_attention", gen_prefill_attention),
                    ("cache_layout", gen_cache_layout), ("scheduler", gen_scheduler),
                    ("prefix_caching", gen_prefix_caching), ("utils", gen_utils)):
        if not only or tag in only:
            fn()
