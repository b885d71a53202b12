"""MI355X-native disaggregated prefill/decode hot path (host side).

The native operators live in ``libsplitwise_hip.so`` (built from ``../csrc``) and are reached
through :mod:`splitwise_amd._C`.  The public names are the reference's (vllm/__init__.py:14-24).
"""
__version__ = "0.1.0"

from .engine.arg_utils import EngineArgs          # noqa: E402,F401
from .engine.async_llm_engine import AsyncLLMEngine   # noqa: E402,F401
from .engine.llm_engine import LLMEngine          # noqa: E402,F401
from .entrypoints.llm import LLM                  # noqa: E402,F401
from .outputs import CompletionOutput, RequestOutput   # noqa: E402,F401
from .sampling_params import SamplingParams       # noqa: E402,F401
