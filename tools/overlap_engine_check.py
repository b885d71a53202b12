#!/usr/bin/env python3
"""Greedy tokens of the one-GPU overlap engine (prompt and decode steps on two streams) against the plain
co-located engine, same seeded model, requests arriving while others decode."""
import random
import sys

sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd"]
import torch  # noqa: E402
from splitwise_amd import EngineArgs, LLMEngine, SamplingParams  # noqa: E402
from splitwise_amd.executor.overlap_executor import OverlapExecutor  # noqa: E402


def run(overlap: bool):
    random.seed(0)
    kw = dict(model="llama-tiny", max_num_seqs=16, max_model_len=512, swap_space=0.01, num_gpu_blocks_override=512,
              pipeline_steps=True, enforce_eager=False)
    if overlap:
        engine = LLMEngine.from_engine_args(EngineArgs(sep_prompt_token=True, **kw), executor_class=OverlapExecutor)
        assert engine.pipeline_split
    else:
        engine = LLMEngine.from_engine_args(EngineArgs(**kw))
    vocab = engine.model_config.get_vocab_size()
    prompts = [[random.randrange(vocab) for _ in range(random.choice([3, 17, 40, 64, 100, 130]))] for _ in range(40)]
    done, step = {}, 0
    pending = list(enumerate(prompts))
    while pending or engine.has_unfinished_requests():
        if pending and step % 2 == 0:
            i, p = pending.pop(0)
            engine.add_request(str(i), None, SamplingParams(temperature=0.0, max_tokens=24 + i % 9, ignore_eos=True),
                               prompt_token_ids=p)
        for out in engine.step():
            if out.finished:
                done[out.request_id] = list(out.outputs[0].token_ids)
        step += 1
        assert step < 5000
    return done


a = run(False)
torch.cuda.empty_cache()
b = run(True)
bad = [k for k in a if a[k] != b.get(k)]
print("requests", len(a), "differing", len(bad))
for k in bad[:5]:
    print(k, a[k], b.get(k))
sys.exit(1 if bad else 0)
