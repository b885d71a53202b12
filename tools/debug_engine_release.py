#!/usr/bin/env python3
"""Does a deleted engine give its device memory back?  Prints what still refers to it when it does not."""
import gc
import sys

sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd"]
import torch  # noqa: E402
from splitwise_amd import EngineArgs, LLMEngine, SamplingParams  # noqa: E402


def build(eager):
    engine = LLMEngine.from_engine_args(EngineArgs(model="llama-tiny", max_num_seqs=16, max_model_len=512,
                                                   swap_space=0.01, num_gpu_blocks_override=4096, pipeline_steps=True,
                                                   enforce_eager=eager, max_context_len_to_capture=256))
    engine.add_request("0", None, SamplingParams(temperature=0.0, max_tokens=8), prompt_token_ids=[1, 2, 3])
    while engine.has_unfinished_requests():
        engine.step()
    return engine


for eager in (True, False):
    free0 = torch.cuda.mem_get_info()[0]
    engine = build(eager)
    used = free0 - torch.cuda.mem_get_info()[0]
    import weakref
    ref = weakref.ref(engine.model_executor.driver_worker)
    del engine
    gc.collect()
    torch.cuda.empty_cache()
    left = free0 - torch.cuda.mem_get_info()[0]
    print(f"eager={eager}: engine took {used >> 20} MiB, {left >> 20} MiB still held, worker alive: {ref() is not None}",
          flush=True)
    if ref() is not None:
        for r in gc.get_referrers(ref())[:6]:
            print("   referrer:", type(r), str(r)[:200])
