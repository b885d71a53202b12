"""Tensor-parallel pools on CPU: PoolExecutor with tensor_parallel_size=4 and sep_prompt_token (a prompt
pool of ranks 0-3 pushing to a token pool of ranks 4-7, the world of vllm/config.py:433-443) drives eight
processes whose workers are CPU stand-ins (tests/split_fakes.py).  Product code under test: process
spawning and the rendezvous store, the real tensor-parallel / stage groups over gloo
(parallel_state.initialize_model_parallel), the per-rank KV-transport handshake (rank i <-> i + tp), the
concurrent prompt-step / decode-step dispatch, sampled tokens reported by the first rank of a pool only,
and the failure paths: a worker that dies must end the engine's wait with an error, never hang it."""
import os
import random
import tempfile
import time

import pytest

import conftest  # noqa: F401  (adds the package and the repo root to sys.path)


def _engine(tp: int, tmp: str, blocks: int = 64, pipeline: bool = False, token_pools: int = 1, staged: bool = False):
    from split_fakes import FakeWorker
    from splitwise_amd import EngineArgs, LLMEngine
    from splitwise_amd.executor.pool_executor import PoolExecutor
    os.environ["SPLIT_FAKE_DIR"] = tmp
    os.environ["SPLIT_FAKE_BLOCKS"] = str(blocks)
    args = EngineArgs(model="llama-tiny", sep_prompt_token=True, tensor_parallel_size=tp, max_num_seqs=6,
                      max_model_len=128, max_num_batched_tokens=512, swap_space=0.001, pipeline_steps=pipeline,
                      token_pools=token_pools, staged_prompt_cache=staged)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    assert pc.world_size == (1 + token_pools) * tp and pc.num_prompt_workers == tp
    executor = PoolExecutor(mc, cc, pc, sc, dc, worker_cls=FakeWorker, devices=[0] * pc.world_size)
    return LLMEngine(mc, cc, pc, sc, dc, executor=executor), executor, cc


@pytest.mark.parametrize("tp,pipeline,token_pools,staged", [(2, False, 1, False), (4, False, 1, False),
                                                            (2, True, 1, False), (2, True, 1, True), (2, True, 2, True)])
def test_split_pools_of_tensor_parallel_ranks(tp, pipeline, token_pools, staged):
    """Lock-stepped pools as in the reference's world; pipelined pools (each pool keeps a step queued: the token
    pool's ranks read their next input tokens from ids their first rank sampled); a prompt pool whose cache holds
    only the prompts in flight; and one prompt pool of two ranks feeding two token pools of two ranks (six
    processes), prompt rank i pushing to rank i of the pool a sequence decodes on."""
    from split_fakes import VOCAB, next_token
    from splitwise_amd import SamplingParams
    with tempfile.TemporaryDirectory() as tmp:
        engine, executor, cc = _engine(tp, tmp, 96 if staged else 64, pipeline, token_pools, staged)
        try:
            assert engine.pipeline_split == pipeline and len(engine.schedulers) == token_pools
            if staged:
                assert cc.num_gpu_blocks == 96 + 16 and executor.prompt_cache_blocks == 76
            else:
                assert cc.num_gpu_blocks == 64                  # the smallest over all ranks (token ranks offer 80)
            assert executor.prompt_pool == list(range(tp)) and executor.token_pool == list(range(tp, 2 * tp))
            assert executor.token_pools[-1] == list(range(token_pools * tp, (token_pools + 1) * tp))
            rng = random.Random(1)
            prompts = {str(i): [rng.randrange(VOCAB) for _ in range(rng.choice([1, 5, 16, 17, 40]))] for i in range(10)}
            lens = {rid: rng.choice([1, 2, 9, 20]) for rid in prompts}
            pending, finished, steps = list(prompts), {}, 0
            while pending or engine.has_unfinished_requests():
                if pending and steps % 2 == 0:
                    rid = pending.pop(0)
                    engine.add_request(rid, None, SamplingParams(temperature=0.0, max_tokens=lens[rid], ignore_eos=True),
                                       prompt_token_ids=prompts[rid])
                for out in engine.step():
                    if out.finished:
                        finished[out.request_id] = list(out.outputs[0].token_ids)
                steps += 1
                assert steps < 2000
            for rid, prompt in prompts.items():
                seq = list(prompt)
                for _ in range(lens[rid]):
                    seq.append(next_token(seq))
                assert finished[rid] == seq[len(prompt):], rid
            assert executor.kv_bytes_pushed() > 0
            for sched in engine.schedulers:
                assert sched.block_manager.gpu_allocator._next_fresh > 0          # every pool served requests
                assert sched.block_manager.get_num_free_gpu_blocks() == cc.num_gpu_blocks
        finally:
            executor.shutdown()


def test_a_colocated_tensor_parallel_pool_runs_pipelined_steps():
    """tensor_parallel_size=2 without split pools, pipeline_steps: step n + 1 is launched on both ranks before
    step n's tokens are known on the host; the second rank never samples and reads its next input tokens from
    what the first rank sampled."""
    from split_fakes import VOCAB, FakeWorker, next_token
    from splitwise_amd import EngineArgs, LLMEngine, SamplingParams
    from splitwise_amd.executor.pool_executor import PoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        os.environ["SPLIT_FAKE_DIR"] = tmp
        os.environ["SPLIT_FAKE_BLOCKS"] = "64"
        args = EngineArgs(model="llama-tiny", tensor_parallel_size=2, max_num_seqs=6, max_model_len=128,
                          max_num_batched_tokens=512, swap_space=0.001, pipeline_steps=True)
        mc, cc, pc, sc, dc = args.create_engine_configs()
        executor = PoolExecutor(mc, cc, pc, sc, dc, worker_cls=FakeWorker, devices=[0, 0])
        try:
            engine = LLMEngine(mc, cc, pc, sc, dc, executor=executor)
            assert engine.pipeline_steps and not engine.pipeline_split
            rng = random.Random(4)
            prompts = {str(i): [rng.randrange(VOCAB) for _ in range(rng.choice([1, 5, 16, 17, 40]))] for i in range(8)}
            lens = {rid: rng.choice([1, 2, 9, 20]) for rid in prompts}
            pending, finished, steps = list(prompts), {}, 0
            while pending or engine.has_unfinished_requests():
                if pending and steps % 2 == 0:
                    rid = pending.pop(0)
                    engine.add_request(rid, None, SamplingParams(temperature=0.0, max_tokens=lens[rid], ignore_eos=True),
                                       prompt_token_ids=prompts[rid])
                for out in engine.step():
                    if out.finished:
                        finished[out.request_id] = list(out.outputs[0].token_ids)
                steps += 1
                assert steps < 2000
            for rid, prompt in prompts.items():
                seq = list(prompt)
                for _ in range(lens[rid]):
                    seq.append(next_token(seq))
                assert finished[rid] == seq[len(prompt):], rid
        finally:
            executor.shutdown()


def test_a_worker_that_dies_in_the_handshake_fails_the_set_up_at_once(monkeypatch):
    monkeypatch.setenv("SPLIT_FAKE_DIE_IN_HANDSHAKE", "3")          # a token-pool rank of the tp=2 world
    monkeypatch.setenv("SPLITWISE_CONTROL_TIMEOUT_S", "60")
    with tempfile.TemporaryDirectory() as tmp:
        began = time.monotonic()
        with pytest.raises(RuntimeError, match="worker 3"):
            _engine(2, tmp)
        assert time.monotonic() - began < 45                       # noticed through the dead process, not the deadline
    import torch.distributed as dist
    if dist.is_initialized():                                      # rank 0 lives in this process
        from splitwise_amd.model_executor import parallel_state
        parallel_state.destroy_model_parallel()
        dist.destroy_process_group()


def _vocab_rank(rank: int, init_method: str, queue) -> None:
    """One rank of a 4-way vocabulary-parallel embedding + output head on gloo (CPU tensors)."""
    try:
        import torch
        import torch.distributed as dist
        from splitwise_amd.model_executor import parallel_state
        from splitwise_amd.model_executor.layers.vocab_parallel_embedding import ParallelLMHead, VocabParallelEmbedding
        from splitwise_amd.model_executor.layers.sampler import Sampler
        dist.init_process_group("gloo", rank=rank, world_size=4, init_method=init_method)
        parallel_state.initialize_model_parallel(4)
        vocab, hidden = 1000, 32                              # padded to 1024: 256 rows per rank, 24 of them padding
        g = torch.Generator().manual_seed(0)
        table = torch.randn(vocab, hidden, generator=g)
        head = torch.randn(vocab, hidden, generator=g)
        emb = VocabParallelEmbedding(vocab, hidden, torch.float32)
        emb.weight_loader(emb.weight, table)
        lm = ParallelLMHead(vocab, hidden, torch.float32)
        lm.weight_loader(lm.weight, head)
        assert emb.weight.shape == (256, hidden) and emb.num_embeddings_padded == 1024
        ids = torch.tensor([0, 255, 256, 511, 767, 768, 999, 3])
        out = emb(ids)
        ok = torch.allclose(out, table[ids], atol=0, rtol=0)
        hid = torch.randn(5, hidden, generator=g)
        logits = Sampler(vocab)._get_logits(hid, lm.weight, None)
        if rank == 0:
            ok = ok and logits.shape == (5, vocab) and torch.allclose(logits, hid @ head.t(), atol=1e-5, rtol=1e-5)
        else:
            ok = ok and logits is None
        dist.barrier()
        dist.destroy_process_group()
        queue.put((rank, "ok" if ok else "wrong values"))
    except BaseException:                         # noqa: BLE001
        import traceback
        queue.put((rank, traceback.format_exc()))


def test_vocabulary_parallel_embedding_and_head_over_four_ranks():
    """vllm/model_executor/layers/vocab_parallel_embedding.py:98-111 and layers/sampler.py:44-57: the lookup is
    exact (each token's row lives on one rank, the others add zeros) and the gathered logits are those of
    the full head, on the pool's first rank only."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    with tempfile.TemporaryDirectory() as tmp:
        init = "file://" + os.path.join(tmp, "rendezvous")
        procs = [ctx.Process(target=_vocab_rank, args=(r, init, queue), daemon=True) for r in range(4)]
        for p in procs:
            p.start()
        try:
            results = dict(queue.get(timeout=120) for _ in range(4))
        finally:
            for p in procs:
                p.join(timeout=10)
                if p.is_alive():
                    p.kill()
    assert results == {0: "ok", 1: "ok", 2: "ok", 3: "ok"}, results
