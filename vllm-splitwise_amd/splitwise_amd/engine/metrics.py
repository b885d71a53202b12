"""Engine statistics: running / waiting / swapped, cache usage, token counters and the TTFT /
TPOT / end-to-end histograms (names of vllm/engine/metrics.py:19-96), logged every few seconds
and exported to Prometheus when prometheus_client is importable."""
import time
from dataclasses import dataclass, field
from typing import List

try:
    from prometheus_client import Counter, Gauge, Histogram
    _PROM = True
except Exception:          # pragma: no cover
    _PROM = False

_METRICS = None


def _metrics():
    global _METRICS
    if _METRICS is None and _PROM:
        _METRICS = dict(
            running=Gauge("vllm:num_requests_running", "Number of requests currently running on GPU."),
            swapped=Gauge("vllm:num_requests_swapped", "Number of requests swapped to CPU."),
            waiting=Gauge("vllm:num_requests_waiting", "Number of requests waiting to be processed."),
            gpu_cache=Gauge("vllm:gpu_cache_usage_perc", "GPU KV-cache usage. 1 means 100 percent usage."),
            prompt_tokens=Counter("vllm:prompt_tokens_total", "Number of prefill tokens processed."),
            generation_tokens=Counter("vllm:generation_tokens_total", "Number of generation tokens processed."),
            ttft=Histogram("vllm:time_to_first_token_seconds", "Histogram of time to first token in seconds.",
                           buckets=[0.001, 0.005, 0.01, 0.02, 0.04, 0.06, 0.08, 0.1, 0.25, 0.5, 0.75, 1.0, 2.5,
                                    5.0, 7.5, 10.0]),
            tpot=Histogram("vllm:time_per_output_token_seconds", "Histogram of time per output token in seconds.",
                           buckets=[0.01, 0.025, 0.05, 0.075, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5, 0.75, 1.0, 2.5]),
            e2e=Histogram("vllm:e2e_request_latency_seconds", "Histogram of end to end request latency in seconds.",
                          buckets=[1.0, 2.5, 5.0, 10.0, 15.0, 20.0, 30.0, 40.0, 50.0, 60.0]),
        )
    return _METRICS


@dataclass
class Stats:
    now: float
    num_running: int
    num_swapped: int
    num_waiting: int
    gpu_cache_usage: float
    cpu_cache_usage: float
    num_prompt_tokens: int
    num_generation_tokens: int
    time_to_first_tokens: List[float] = field(default_factory=list)
    time_per_output_tokens: List[float] = field(default_factory=list)
    time_e2e_requests: List[float] = field(default_factory=list)


class StatLogger:
    def __init__(self, local_interval: float) -> None:
        self.local_interval = local_interval
        self.last_local_log = time.monotonic()
        self.num_prompt_tokens: List[int] = []
        self.num_generation_tokens: List[int] = []

    def log(self, stats: Stats) -> None:
        m = _metrics()
        if m is not None:
            m["running"].set(stats.num_running)
            m["swapped"].set(stats.num_swapped)
            m["waiting"].set(stats.num_waiting)
            m["gpu_cache"].set(stats.gpu_cache_usage)
            m["prompt_tokens"].inc(stats.num_prompt_tokens)
            m["generation_tokens"].inc(stats.num_generation_tokens)
            for v in stats.time_to_first_tokens:
                m["ttft"].observe(v)
            for v in stats.time_per_output_tokens:
                m["tpot"].observe(v)
            for v in stats.time_e2e_requests:
                m["e2e"].observe(v)
        self.num_prompt_tokens.append(stats.num_prompt_tokens)
        self.num_generation_tokens.append(stats.num_generation_tokens)
        now = time.monotonic()
        if now - self.last_local_log > self.local_interval:
            dt = now - self.last_local_log
            print(f"Avg prompt throughput: {sum(self.num_prompt_tokens) / dt:.1f} tokens/s, "
                  f"Avg generation throughput: {sum(self.num_generation_tokens) / dt:.1f} tokens/s, "
                  f"Running: {stats.num_running} reqs, Swapped: {stats.num_swapped} reqs, "
                  f"Pending: {stats.num_waiting} reqs, GPU KV cache usage: {stats.gpu_cache_usage * 100:.1f}%",
                  flush=True)
            self.num_prompt_tokens, self.num_generation_tokens = [], []
            self.last_local_log = now
