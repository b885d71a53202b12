#!/usr/bin/env python3
"""Host time of an engine step at small batch: cProfile over decode steps of a few sequences (the GPU step is
shorter than the host step there, so the host sets the token rate)."""
import cProfile
import pstats
import sys
import time

sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd"]
import torch  # noqa: E402
from splitwise_amd import EngineArgs, LLMEngine, SamplingParams  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    pipeline = len(sys.argv) > 2 and sys.argv[2] == "pipeline"
    engine = LLMEngine.from_engine_args(EngineArgs(model="llama-2-7b", dtype="bfloat16", max_num_seqs=256, max_model_len=2048,
                                                   num_gpu_blocks_override=4096, swap_space=0.25,
                                                   max_context_len_to_capture=1024, pipeline_steps=pipeline))
    params = SamplingParams(temperature=0.0, max_tokens=400, ignore_eos=True)
    for i in range(n):
        engine.add_request(str(i), None, params, prompt_token_ids=list(range(100 + i, 200 + i)))
    for _ in range(30):
        engine.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prof = cProfile.Profile()
    prof.enable()
    steps = 200
    for _ in range(steps):
        engine.step()
    prof.disable()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n} sequences, pipeline={pipeline}: {dt / steps * 1e3:.2f} ms per step")
    pstats.Stats(prof).sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
