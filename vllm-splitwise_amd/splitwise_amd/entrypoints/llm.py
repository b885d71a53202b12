"""Offline inference front-end (vllm/entrypoints/llm.py:16-210): ``LLM(model, ...).generate``."""
from typing import List, Optional, Union

from ..engine.arg_utils import EngineArgs
from ..engine.llm_engine import LLMEngine
from ..outputs import RequestOutput
from ..sampling_params import SamplingParams


class LLM:
    def __init__(self, model: str, tokenizer=None, tensor_parallel_size: int = 1, dtype: str = "auto",
                 quantization: Optional[str] = None, seed: int = 0, gpu_memory_utilization: float = 0.9,
                 swap_space: int = 4, enforce_eager: bool = False, max_context_len_to_capture: Optional[int] = None,
                 **kwargs) -> None:
        engine_args = EngineArgs(model=model, tensor_parallel_size=tensor_parallel_size, dtype=dtype,
                                 quantization=quantization, seed=seed,
                                 gpu_memory_utilization=gpu_memory_utilization, swap_space=swap_space,
                                 enforce_eager=enforce_eager,
                                 max_context_len_to_capture=max_context_len_to_capture, **kwargs)
        self.llm_engine = LLMEngine.from_engine_args(engine_args, tokenizer=tokenizer)
        self.request_counter = 0

    def generate(self, prompts: Optional[Union[str, List[str]]] = None,
                 sampling_params: Optional[SamplingParams] = None,
                 prompt_token_ids: Optional[List[List[int]]] = None) -> List[RequestOutput]:
        if prompts is None and prompt_token_ids is None:
            raise ValueError("Either prompts or prompt_token_ids must be provided.")
        if isinstance(prompts, str):
            prompts = [prompts]
        if prompts is not None and prompt_token_ids is not None and len(prompts) != len(prompt_token_ids):
            raise ValueError("The lengths of prompts and prompt_token_ids must be the same.")
        if sampling_params is None:
            sampling_params = SamplingParams()
        n = len(prompts) if prompts is not None else len(prompt_token_ids)
        for i in range(n):
            self.llm_engine.add_request(str(self.request_counter), prompts[i] if prompts is not None else None,
                                        sampling_params,
                                        None if prompt_token_ids is None else prompt_token_ids[i])
            self.request_counter += 1
        return self._run_engine()

    def _run_engine(self) -> List[RequestOutput]:
        done: List[RequestOutput] = []
        while self.llm_engine.has_unfinished_requests():
            for out in self.llm_engine.step():
                if out.finished:
                    done.append(out)
        return sorted(done, key=lambda x: int(x.request_id))
