"""The N>1 host path on CPU, world_size 2 over gloo: the engine with the split executor on rank 0
and the token worker's command loop on rank 1, both with CPU stand-in workers (tests/split_fakes.py).
Everything else is the product code: scheduler split scheduling and hand-off epochs, the executor's
overlapped dispatch, the object channel, the worker loop and the transport handshake."""
import os
import random
import tempfile

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp


def _rank_main(rank: int, init_method: str, tmpdir: str, result_path: str, blocks: int, pipeline: bool,
               token_pools: int = 1, staged: bool = False) -> None:
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import conftest  # noqa: F401  (adds the package and the repo root to sys.path)
    os.environ["SPLIT_FAKE_DIR"] = tmpdir
    os.environ["SPLIT_FAKE_BLOCKS"] = str(blocks)
    dist.init_process_group("gloo", init_method=init_method, rank=rank, world_size=1 + token_pools)
    from split_fakes import VOCAB, FakeWorker, next_token
    from splitwise_amd import EngineArgs, LLMEngine, SamplingParams
    from splitwise_amd.executor.channel import DistChannel
    from splitwise_amd.executor.remote_worker import worker_loop
    from splitwise_amd.executor.splitwise_executor import SplitwiseExecutor

    args = EngineArgs(model="llama-tiny", sep_prompt_token=True, max_num_seqs=6, max_model_len=128,
                      max_num_batched_tokens=512, swap_space=0.001, pipeline_steps=pipeline,
                      token_pools=token_pools, staged_prompt_cache=staged)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    assert pc.world_size == 1 + token_pools and pc.staged_prompt_cache == (staged or token_pools > 1)
    if rank >= 1:
        cfg = {"model_config": mc, "parallel_config": pc, "scheduler_config": sc, "cache_config": cc, "rank": rank,
               "local_rank": 0, "transport": "kernel"}
        worker_loop(DistChannel(0), cfg, worker_cls=FakeWorker)
        dist.barrier()
        return
    executor = SplitwiseExecutor(mc, cc, pc, sc, dc, channels=[DistChannel(1 + j) for j in range(token_pools)],
                                 worker_cls=FakeWorker)
    if pc.staged_prompt_cache:
        # the token pools' size (they offer 16 blocks more than the prompt side), and a prompt cache of two
        # prompt steps' worth: 512 tokens in at most 6 prompts -> 2 * (32 + 6)
        assert cc.num_gpu_blocks == blocks + 16 and executor.prompt_cache_blocks == 76
        blocks = (blocks + 16) * token_pools
    else:
        assert cc.num_gpu_blocks == blocks                       # the smaller of the two pools
    engine = LLMEngine(mc, cc, pc, sc, dc, executor=executor)
    assert engine.pipeline_split == pipeline and len(engine.schedulers) == token_pools
    rng = random.Random(0)
    prompts = {str(i): [rng.randrange(VOCAB) for _ in range(rng.choice([1, 5, 16, 17, 40, 60]))] for i in range(14)}
    lens = {rid: rng.choice([1, 2, 9, 30, 40]) for rid in prompts}

    def generate(prompt, n):
        seq = list(prompt)
        for _ in range(n):
            seq.append(next_token(seq))
        return seq[len(prompt):]

    # every third request ends on a stop token (its own 4th token, or an earlier equal one)
    stops = {rid: [generate(prompts[rid], lens[rid])[min(3, lens[rid] - 1)]] for rid in prompts if int(rid) % 3 == 0}
    pending = list(prompts)
    finished = {}
    steps = 0
    while pending or engine.has_unfinished_requests():
        if pending and steps % 2 == 0:                           # requests keep arriving while others decode
            rid = pending.pop(0)
            engine.add_request(rid, None, SamplingParams(temperature=0.0, max_tokens=lens[rid], ignore_eos=True,
                                                         stop_token_ids=stops.get(rid, [])),
                               prompt_token_ids=prompts[rid])
        for out in engine.step():
            if out.finished:
                finished[out.request_id] = list(out.outputs[0].token_ids)
        steps += 1
        assert steps < 2000
    expected = {}
    for rid, prompt in prompts.items():
        out = generate(prompt, lens[rid])
        if rid in stops:
            out = out[:out.index(stops[rid][0]) + 1]
        expected[rid] = out
    pushed = executor.kv_bytes_pushed()
    preempted = getattr(engine.scheduler, "num_preemptions", 0)
    executor.shutdown()
    free = engine.scheduler.block_manager.get_num_free_gpu_blocks()
    verdict = "ok" if finished == expected and pushed > 0 else f"mismatch: {finished} != {expected} pushed={pushed}"
    if verdict == "ok" and free != blocks:
        verdict = f"{blocks - free} cache blocks were never given back"
    if blocks < 64 and preempted == 0:
        verdict = "the small cache did not force a single preemption: the scenario tests nothing"
    if token_pools > 1 and verdict == "ok":
        used = [sched.block_manager.gpu_allocator._next_fresh for sched in engine.schedulers]
        if min(used) == 0:
            verdict = f"a token pool never received a request: fresh blocks used per pool {used}"
    with open(result_path, "w") as f:
        f.write(verdict)
    dist.barrier()


@pytest.mark.parametrize("pipeline", [False, True])  # True: both pools keep a step in flight (placeholder tokens)
@pytest.mark.parametrize("blocks", [64, 12])       # 12 blocks: decodes run out of cache, sequences are
def test_split_engine_over_gloo_world_size_2(blocks, pipeline):  # preempted, recomputed and handed over again
    with tempfile.TemporaryDirectory() as tmp:
        result = os.path.join(tmp, "result.txt")
        # rendezvous through a file: no port to pick before the ranks exist
        mp.spawn(_rank_main, args=("file://" + os.path.join(tmp, "rendezvous"), tmp, result, blocks, pipeline),
                 nprocs=2, join=True)
        assert open(result).read() == "ok"


@pytest.mark.parametrize("token_pools,staged", [(1, True), (2, True)])
def test_one_prompt_worker_feeds_token_pools_with_their_own_block_spaces(token_pools, staged):
    """World size 2 with a staged prompt cache, and world size 3: one prompt worker (rank 0, with the engine)
    feeding two token workers (ranks 1 and 2).  The prompt side numbers its few blocks by itself and forgets
    them after every step; each token pool has its own scheduler, block space and hand-off slots.  Every token
    is computed from what the token worker's cache holds, so a push to the wrong pool, a wrong block mapping
    or a wrong epoch shows up as a wrong token or a failed wait."""
    with tempfile.TemporaryDirectory() as tmp:
        result = os.path.join(tmp, "result.txt")
        mp.spawn(_rank_main, args=("file://" + os.path.join(tmp, "rendezvous"), tmp, result, 80, True, token_pools,
                                   staged), nprocs=1 + token_pools, join=True)
        assert open(result).read() == "ok"


def test_block_pair_runs_need_both_sides_consecutive():
    import conftest  # noqa: F401
    from splitwise_amd.utils import coalesce_block_pairs
    assert coalesce_block_pairs([0, 1, 2, 3], [7, 8, 9, 10]) == [(0, 7, 4)]
    assert coalesce_block_pairs([0, 1, 2, 3], [7, 8, 3, 4]) == [(0, 7, 2), (2, 3, 2)]
    assert coalesce_block_pairs([4, 5, 9], [1, 2, 3]) == [(4, 1, 2), (9, 3, 1)]
    assert coalesce_block_pairs([], []) == []


def test_slab_layout_is_symmetric_and_avoids_the_unmappable_sizes():
    import conftest  # noqa: F401
    from splitwise_amd.worker.kv_comm import FLAG_BYTES, _ipc_safe_size, slab_layout
    offsets, flags, total = slab_layout(32, 128 * 1024, 9500)
    assert offsets == slab_layout(32, 128 * 1024, 9500)[0] and flags == total - FLAG_BYTES
    assert offsets[0] == (0, 9500 * 128 * 1024) and offsets[1][0] == 2 * 9500 * 128 * 1024
    for gib in (0.5, 1.0, 1.99, 2.01, 3.9, 6, 19, 23, 42.1, 64):
        n = _ipc_safe_size(int(gib * 2 ** 30))
        assert n >= int(gib * 2 ** 30) and n % 2 ** 32 < 2 ** 31
    assert _ipc_safe_size(1 << 30) == 1 << 30


def test_decode_commands_travel_without_token_ids():
    """What crosses the control channel for a decode step: (length, last token) per sequence and
    compact sample tuples back; prompt steps keep every token."""
    import pickle
    from splitwise_amd.sampling_params import SamplingParams
    from splitwise_amd.sequence import (Logprob, SamplerOutput, SequenceData, SequenceGroupMetadata,
                                        SequenceGroupOutput, SequenceOutput, pack_sampler_output,
                                        slim_decode_metadata, unpack_sampler_output)
    sp = SamplingParams(temperature=0.0, max_tokens=4)
    data = SequenceData(list(range(700)), [5, 6, 7])
    decode = [SequenceGroupMetadata("a", False, {3: data}, sp, {3: [9, 4]})]
    slim = slim_decode_metadata(decode)
    d = slim[0].seq_data[3]
    assert (d.get_len(), d.get_last_token_id()) == (703, 7) and slim[0].block_tables == {3: [9, 4]}
    assert not slim[0].is_prompt and slim[0].request_id == "a" and slim[0].sampling_params is sp
    assert len(pickle.dumps(slim)) < len(pickle.dumps(decode)) // 3
    with pytest.raises(RuntimeError):
        d.get_token_ids()
    prompt = [SequenceGroupMetadata("b", True, {4: data}, sp, {4: [1]})]
    assert slim_decode_metadata(prompt) is prompt
    out = SamplerOutput([SequenceGroupOutput([SequenceOutput(3, 11, {11: Logprob(-0.25)})], None),
                         SequenceGroupOutput([SequenceOutput(8, 2, {2: Logprob(0.0)})], None)])
    back = unpack_sampler_output(pickle.loads(pickle.dumps(pack_sampler_output(out))))
    assert [g.samples[0] for g in back] == [g.samples[0] for g in out]
