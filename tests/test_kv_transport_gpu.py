"""GPU parity: the KV transport (push / signal / wait through the C ABI) and the split
prompt-pool / token-pool engine.

Known-answer pattern of the reference's transport test (tests/distributed/test_kvcache_comm.py
:28-42 with vllm/worker/worker.py:350-386): the sender's cache holds a constant per (rank,
layer, tensor); after put + signal / wait the receiver must hold the sender's constants.  Here
it is also checked for arbitrary (block_start, num_blocks) runs: bytes inside the runs equal
the source, everything else is untouched (bit-exact).  Both workers share the one GPU of the
test box; the IPC mapping, the kernels and the flag protocol are the same as across two GPUs.
"""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_push_runs_bit_exact_and_rest_untouched():
    from splitwise_amd import _C
    torch.manual_seed(0)
    nbytes = 3 * 1024 * 1024 + 4096
    src = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda")
    dst = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda")
    before = dst.clone()
    # ragged runs: tiny, odd multiples of 16, larger than one 64-KiB chunk, adjacent runs
    runs = [(0, 16), (4096, 48), (65536, 65536 * 3 + 32), (1048576, 16), (1048592, 131072), (nbytes - 160, 160)]
    _C.kv_ops.kv_push(dst.data_ptr(), src.data_ptr(), [o for o, _ in runs], [n for _, n in runs],
                      torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    expect = before.clone()
    for off, n in runs:
        expect[off:off + n] = src[off:off + n]
    assert torch.equal(dst, expect)
    # more runs than one launch carries
    many = [(i * 4096, 16 * (1 + i % 5)) for i in range(130)]
    dst2 = before.clone()
    _C.kv_ops.kv_push(dst2.data_ptr(), src.data_ptr(), [o for o, _ in many], [n for _, n in many],
                      torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    expect = before.clone()
    for off, n in many:
        expect[off:off + n] = src[off:off + n]
    assert torch.equal(dst2, expect)
    # a capped launch (the form used over an xGMI link): 3 workgroups walk all 58 chunks of the runs
    dst3 = before.clone()
    _C.kv_ops.kv_push(dst3.data_ptr(), src.data_ptr(), [o for o, _ in runs], [n for _, n in runs],
                      torch.cuda.current_stream().cuda_stream, 3)
    _C.kv_ops.kv_push(dst3.data_ptr(), src.data_ptr(), [o for o, _ in many], [n for _, n in many],
                      torch.cuda.current_stream().cuda_stream, 1)
    torch.cuda.synchronize()
    for off, n in runs:
        expect[off:off + n] = src[off:off + n]
    assert torch.equal(dst3, expect)
    with pytest.raises(RuntimeError, match="16-byte"):
        _C.kv_ops.kv_push(dst.data_ptr(), src.data_ptr(), [8], [16], 0)
    _C.kv_ops.kv_push(dst.data_ptr(), src.data_ptr(), [], [], 0)          # empty: no-op


def test_mapped_push_moves_every_range_to_its_own_destination():
    """sw_kv_push_mapped: a prompt worker that numbers its blocks by itself pushes source range i to
    destination range i; bytes inside the destination ranges equal their sources, everything else is untouched."""
    from splitwise_amd import _C
    torch.manual_seed(1)
    src = torch.randint(0, 256, (2 * 1024 * 1024,), dtype=torch.uint8, device="cuda")
    dst = torch.randint(0, 256, (5 * 1024 * 1024 + 48,), dtype=torch.uint8, device="cuda")
    before = dst.clone()
    # (source offset, destination offset, bytes): tiny, several chunks, adjacent sources to distant destinations
    runs = [(0, 4 * 1024 * 1024, 16), (16, 32, 65536 * 2 + 48), (1048576, 3 * 1024 * 1024, 65536),
            (1048576 + 65536, 1024 * 1024, 131072 + 16), (2 * 1024 * 1024 - 160, 5 * 1024 * 1024 - 112, 160)]
    stream = torch.cuda.current_stream().cuda_stream
    for cap in (0, 2):
        out = before.clone()
        _C.kv_ops.kv_push_mapped(out.data_ptr(), src.data_ptr(), [d for _, d, _ in runs], [s_ for s_, _, _ in runs],
                                 [n for _, _, n in runs], stream, cap)
        torch.cuda.synchronize()
        expect = before.clone()
        for s_, d, n in runs:
            expect[d:d + n] = src[s_:s_ + n]
        assert torch.equal(out, expect), cap
    many = [(16 * i, 4096 * i, 16) for i in range(120)]              # more ranges than one launch carries
    out = before.clone()
    _C.kv_ops.kv_push_mapped(out.data_ptr(), src.data_ptr(), [d for _, d, _ in many], [s_ for s_, _, _ in many],
                             [n for _, _, n in many], stream)
    torch.cuda.synchronize()
    expect = before.clone()
    for s_, d, n in many:
        expect[d:d + n] = src[s_:s_ + n]
    assert torch.equal(out, expect)
    with pytest.raises(RuntimeError, match="16-byte"):
        _C.kv_ops.kv_push_mapped(dst.data_ptr(), src.data_ptr(), [16], [8], [16], 0)
    with pytest.raises(RuntimeError, match="differ in length"):
        _C.kv_ops.kv_push_mapped(dst.data_ptr(), src.data_ptr(), [16, 32], [16], [16, 16], 0)


def test_signal_then_wait_and_timeout():
    from splitwise_amd import _C
    flags = torch.zeros(260, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    status_ptr = flags.data_ptr() + 4 * 256
    _C.kv_ops.kv_signal(flags.data_ptr(), 7, 1, stream)
    _C.kv_ops.kv_wait(flags.data_ptr(), 7, 1, 1000, status_ptr, stream)
    torch.cuda.synchronize()
    assert flags[7].item() == 1 and flags[256].item() == 0
    # epochs only grow: a later epoch satisfies an earlier wait
    _C.kv_ops.kv_signal(flags.data_ptr(), 7, 3, stream)
    _C.kv_ops.kv_wait(flags.data_ptr(), 7, 2, 1000, status_ptr, stream)
    torch.cuda.synchronize()
    assert flags[256].item() == 0
    # nobody signals slot 9: the bounded wait gives up and reports the slot
    _C.kv_ops.kv_wait(flags.data_ptr(), 9, 1, 2000, status_ptr, stream)
    torch.cuda.synchronize()
    assert flags[256].item() == 10


def _split_llm(transport, **kw):
    from splitwise_amd import LLM
    return LLM("llama-tiny", sep_prompt_token=True, enforce_eager=False, num_gpu_blocks_override=192,
               max_num_seqs=8, max_model_len=512, swap_space=0.01, **kw)


@pytest.mark.parametrize("transport,blocks,kv_dtype", [("kernel", 64, "auto"), ("runtime", 64, "auto"),
                                                       ("kernel", 16, "fp8_e5m2"), ("runtime", 16, "fp8_e5m2")])
def test_transport_self_test_between_two_processes(transport, blocks, kv_dtype):
    """set_gpucache / send_recv_kvcache_all / check_gpucache of the reference, both transports.
    The 16-block fp8 cache is 132 KiB: it lives inside a shared segment of the caching allocator,
    at an offset from the start of the device allocation that the peer maps."""
    from splitwise_amd.engine.arg_utils import EngineArgs
    from splitwise_amd.executor.splitwise_executor import SplitwiseExecutor
    args = EngineArgs(model="llama-tiny", sep_prompt_token=True, enforce_eager=True, num_gpu_blocks_override=blocks,
                      max_num_seqs=4, max_model_len=256, swap_space=0.01, kv_cache_dtype=kv_dtype)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    ex = SplitwiseExecutor(mc, cc, pc, sc, dc, num_gpu_blocks_override=blocks, prompt_device=0, token_device=0,
                           transport=transport)
    try:
        ex._remote_call("set_gpucache")
        ex.driver_worker.set_gpucache()
        ex._remote_async("send_recv_kvcache_all")
        ex.driver_worker.send_recv_kvcache_all()
        ex._remote_wait()
        ex._remote_call("check_gpucache")          # raises in the token worker on a mismatch
        ex.driver_worker.check_gpucache()
        elt = 1 if kv_dtype == "fp8_e5m2" else 2
        assert ex.kv_bytes_pushed() == 2 * mc.hf_config["num_hidden_layers"] * blocks * 16 * 2 * 64 * elt
        slab = ex.driver_worker.cache_engine.gpu_slab
        if blocks == 16:
            from splitwise_amd import _C
            assert _C.kv_ops.mem_address_range(slab.data_ptr())[1] > slab.numel()     # the case this test is for
    finally:
        ex.shutdown()


def test_push_workgroup_cap_is_measured_at_set_up(monkeypatch):
    """Over a link the workgroup cap of a push launch is picked by measurement when the transport is set up
    (here forced for a peer on the same device): every candidate is timed, the smallest one within 8 % of the
    best is kept, the bytes counter does not see the trial pushes, and the reference's self-test still finds
    every block on the other side."""
    from splitwise_amd.engine.arg_utils import EngineArgs
    from splitwise_amd.executor.splitwise_executor import SplitwiseExecutor
    monkeypatch.setenv("SPLITWISE_KV_PUSH_CALIBRATE", "1")
    monkeypatch.delenv("SPLITWISE_KV_PUSH_WORKGROUPS", raising=False)
    blocks = 64
    args = EngineArgs(model="llama-tiny", sep_prompt_token=True, enforce_eager=True, num_gpu_blocks_override=blocks,
                      max_num_seqs=4, max_model_len=256, swap_space=0.01)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    ex = SplitwiseExecutor(mc, cc, pc, sc, dc, num_gpu_blocks_override=blocks, prompt_device=0, token_device=0,
                           transport="kernel")
    try:
        comm = ex.driver_worker.kvcache_comm_manager.kvcache_comm
        table = comm.push_calibration[0]
        assert sorted(table) == [16, 32, 64, 128, 256] and all(v > 0 for v in table.values())
        chosen = comm.push_workgroups[0]
        assert chosen in table and table[chosen] >= max(table.values()) / 1.08
        assert all(table[c] < max(table.values()) / 1.08 for c in table if c < chosen)
        assert ex.kv_bytes_pushed() == 0
        ex._remote_call("set_gpucache")
        ex.driver_worker.set_gpucache()
        ex._remote_async("send_recv_kvcache_all")
        ex.driver_worker.send_recv_kvcache_all()
        ex._remote_wait()
        ex._remote_call("check_gpucache")
        ex.driver_worker.check_gpucache()
        assert ex.kv_bytes_pushed() == 2 * mc.hf_config["num_hidden_layers"] * blocks * 16 * 2 * 64 * 2
    finally:
        ex.shutdown()


def test_a_transport_that_does_not_deliver_fails_at_set_up(monkeypatch):
    """Before the first hand-off a few blocks with known bytes travel the real path and the token worker compares
    what its cache then holds.  Here the prompt side's push is made to do nothing: the flag still arrives, the
    bytes do not, and the executor's construction fails instead of a later decode step running on blocks that
    never came."""
    from splitwise_amd import _C
    from splitwise_amd.engine.arg_utils import EngineArgs
    from splitwise_amd.executor.splitwise_executor import SplitwiseExecutor
    monkeypatch.setattr(_C.kv_ops, "kv_push_mapped", lambda *a, **k: None)
    args = EngineArgs(model="llama-tiny", sep_prompt_token=True, enforce_eager=True, num_gpu_blocks_override=32,
                      max_num_seqs=4, max_model_len=256, swap_space=0.01)
    mc, cc, pc, sc, dc = args.create_engine_configs()
    with pytest.raises(RuntimeError, match="set-up check did not arrive intact"):
        SplitwiseExecutor(mc, cc, pc, sc, dc, num_gpu_blocks_override=32, prompt_device=0, token_device=0,
                          transport="kernel")


@pytest.mark.parametrize("overlap", [True, False, "pipelined"])
def test_split_engine_generates_like_the_cpu_model(overlap):
    """1 prompt worker + 1 token worker (two processes): every decode step runs on KV that was
    pushed from the prompt worker; greedy tokens must agree with the whole-model CPU oracle."""
    from splitwise_amd import SamplingParams
    from oracle import model as omodel
    random.seed(0)
    llm = _split_llm("kernel", pipeline_steps=overlap == "pipelined")
    try:
        assert llm.llm_engine.pipeline_split == (overlap == "pipelined")
        llm.llm_engine.overlap_pools = bool(overlap)
        prompts = [[random.randrange(512) for _ in range(n)] for n in (5, 33, 64, 17, 100, 1, 16, 48, 7, 90, 31)]
        outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0.0, max_tokens=20))
        assert all(len(o.outputs[0].token_ids) == 20 for o in outs)
        runner = llm.llm_engine.model_executor.driver_worker.model_runner
        sd = {k: v.detach().cpu() for k, v in runner.model.state_dict().items()}
        cfg = runner.model_config.hf_config
        for prompt, out in zip(prompts, outs):
            gen = out.outputs[0].token_ids
            ref = omodel.logits("llama", sd, cfg, prompt + gen)
            for j, tok in enumerate(gen):
                row = ref[len(prompt) + j - 1]
                assert row[tok] >= row.max() - 3e-2, (j, tok, int(row.argmax()))
        pushed = llm.llm_engine.model_executor.kv_bytes_pushed()
        blocks = sum(-(-len(p) // 16) for p in prompts)
        assert pushed == blocks * 2 * cfg["num_hidden_layers"] * 16 * 2 * 64 * 2      # K+V, bf16... f16 blocks
        # the side that waits is the token worker: its status word is the one that can be set
        llm.llm_engine.model_executor._remote_call("check_kv_status")
    finally:
        llm.llm_engine.model_executor.shutdown()


def _tokens_agree_with_cpu_model(llm, prompts, outs):
    from oracle import model as omodel
    runner = llm.llm_engine.model_executor.driver_worker.model_runner
    sd = {k: v.detach().cpu() for k, v in runner.model.state_dict().items()}
    cfg = runner.model_config.hf_config
    for prompt, out in zip(prompts, outs):
        gen = out.outputs[0].token_ids
        ref = omodel.logits("llama", sd, cfg, prompt + gen)
        for j, tok in enumerate(gen):
            row = ref[len(prompt) + j - 1]
            assert row[tok] >= row.max() - 3e-2, (j, tok, int(row.argmax()))
    return cfg


@pytest.mark.parametrize("token_pools", [1, 2])
def test_staged_prompt_cache_and_two_token_pools_generate_like_the_cpu_model(token_pools):
    """The prompt worker keeps only the prompts in flight under block numbers of its own (a few blocks where
    the token workers have 192), and with two token pools feeds two token workers (three processes on the one
    GPU), each with its own block space and hand-off slots: every decode step still runs on exactly the KV the
    prompt worker computed, so greedy tokens must agree with the whole-model CPU oracle."""
    from splitwise_amd import SamplingParams
    random.seed(2)
    llm = _split_llm("kernel", pipeline_steps=True, staged_prompt_cache=True, token_pools=token_pools,
                     max_num_batched_tokens=512)
    try:
        engine = llm.llm_engine
        ex = engine.model_executor
        assert engine.pipeline_split and len(engine.schedulers) == token_pools
        assert ex.prompt_cache_blocks == 2 * (512 // 16 + 8)
        assert ex.driver_worker.gpu_cache[0][0].shape[0] == ex.prompt_cache_blocks
        prompts = [[random.randrange(512) for _ in range(n)] for n in (5, 33, 64, 17, 100, 1, 16, 48, 7, 90, 31, 65)]
        outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0.0, max_tokens=20))
        assert all(len(o.outputs[0].token_ids) == 20 for o in outs)
        cfg = _tokens_agree_with_cpu_model(llm, prompts, outs)
        blocks = sum(-(-len(p) // 16) for p in prompts)
        assert ex.kv_bytes_pushed() == blocks * 2 * cfg["num_hidden_layers"] * 16 * 2 * 64 * 2
        for sched in engine.schedulers:
            assert sched.block_manager.gpu_allocator._next_fresh > 0           # every pool served requests
            assert sched.block_manager.get_num_free_gpu_blocks() == 192        # ... and got every block back
        for lane in range(token_pools):
            ex._remote_call("check_kv_status", lane)
    finally:
        llm.llm_engine.model_executor.shutdown()


def test_blocks_of_prompts_that_end_at_their_first_token_are_reused_safely():
    """Requests that finish at their prompt step (max_tokens=1) give their token-side blocks back the moment the
    step is reported finished, and with a small cache the very next decode step hands them to a running
    sequence.  The pushes of that prompt step run on the transport's own stream: the step must not be reported
    finished before they have left, or a late push lands on the new owner's KV.  Tokens of the long sequences
    must still agree with the CPU model."""
    from splitwise_amd import LLM, SamplingParams
    random.seed(3)
    llm = LLM("llama-tiny", sep_prompt_token=True, enforce_eager=False, num_gpu_blocks_override=40, max_num_seqs=8,
              max_model_len=512, swap_space=0.01, pipeline_steps=True)
    try:
        engine = llm.llm_engine
        long_prompts = [[random.randrange(512) for _ in range(n)] for n in (30, 45, 17)]
        short = [[random.randrange(512) for _ in range(random.choice([60, 90, 120]))] for _ in range(24)]
        params_long = SamplingParams(temperature=0.0, max_tokens=70, ignore_eos=True)
        params_one = SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True)
        for i, p in enumerate(long_prompts):
            engine.add_request(f"long{i}", None, params_long, prompt_token_ids=p)
        done, fed, steps = {}, 0, 0
        while engine.has_unfinished_requests() or fed < len(short):
            if fed < len(short) and steps % 2 == 0:                  # one-token requests keep arriving
                engine.add_request(f"one{fed}", None, params_one, prompt_token_ids=short[fed])
                fed += 1
            for out in engine.step():
                if out.finished:
                    done[out.request_id] = out
            steps += 1
            assert steps < 3000
        assert len(done) == len(long_prompts) + len(short)
        outs = [done[f"long{i}"] for i in range(len(long_prompts))]
        assert all(len(o.outputs[0].token_ids) == 70 for o in outs)
        _tokens_agree_with_cpu_model(llm, long_prompts, outs)
        ones = [done[f"one{i}"] for i in range(len(short))]
        _tokens_agree_with_cpu_model(llm, short, ones)
    finally:
        llm.llm_engine.model_executor.shutdown()


def test_a_hand_off_signal_that_never_comes_fails_the_decode_step(monkeypatch):
    """The prompt worker pushes the KV blocks but its signal is withheld: the token worker's bounded wait
    runs out, and the engine must get a RuntimeError for that decode step instead of tokens computed on
    blocks that were never announced (the reference's proxy_channel.wait(1e8) traps, kv_comm_kernels.cu:7)."""
    from splitwise_amd import SamplingParams
    monkeypatch.setenv("SPLITWISE_KV_WAIT_SPINS", "20000")          # the worker process inherits it
    llm = _split_llm("kernel")
    try:
        comm = llm.llm_engine.model_executor.driver_worker.kvcache_comm_manager
        monkeypatch.setattr(comm, "signal_and_flush", lambda *a, **k: None)
        with pytest.raises(RuntimeError, match="hand-off wait timed out"):
            llm.generate(prompt_token_ids=[[1, 2, 3, 4, 5]], sampling_params=SamplingParams(temperature=0.0, max_tokens=4))
    finally:
        llm.llm_engine.model_executor.shutdown()
