"""Engine statistics (vllm/engine/metrics.py:18-230): per step the engine hands a ``Stats`` record to
``StatLogger``, which feeds the Prometheus series (same names, help texts, buckets and the
``model_name`` label as the reference, so its dashboards read this server unchanged) and prints a
throughput line every ``local_interval`` seconds.

Series: gauges ``vllm:num_requests_{running,swapped,waiting}``, ``vllm:{gpu,cpu}_cache_usage_perc``,
``vllm:avg_{prompt,generation}_throughput_toks_per_s`` (legacy, per logging interval); counters
``vllm:{prompt,generation}_tokens_total``; histograms ``vllm:time_to_first_token_seconds``,
``vllm:time_per_output_token_seconds``, ``vllm:e2e_request_latency_seconds``; info
``vllm:cache_config``.  Without ``prometheus_client`` only the printed line remains.
"""
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional

try:
    from prometheus_client import REGISTRY, Counter, Gauge, Histogram, Info, disable_created_metrics
    disable_created_metrics()
    _PROM = True
except Exception:          # pragma: no cover
    _PROM = False

_TTFT_BUCKETS = [0.001, 0.005, 0.01, 0.02, 0.04, 0.06, 0.08, 0.1, 0.25, 0.5, 0.75, 1.0, 2.5, 5.0, 7.5, 10.0]
_TPOT_BUCKETS = [0.01, 0.025, 0.05, 0.075, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5, 0.75, 1.0, 2.5]
_E2E_BUCKETS = [1.0, 2.5, 5.0, 10.0, 15.0, 20.0, 30.0, 40.0, 50.0, 60.0]


class Metrics:
    """The Prometheus collectors.  A process has one live set: creating another (a second engine, a
    test) first unregisters the previous ``vllm:*`` collectors, as the reference does (metrics.py:22-25)."""

    def __init__(self, labelnames: List[str]) -> None:
        for collector in list(REGISTRY._collector_to_names):
            if "vllm" in getattr(collector, "_name", ""):
                REGISTRY.unregister(collector)
        self.info_cache_config = Info("vllm:cache_config", "information of cache_config")

        def gauge(name, doc):
            return Gauge(name, doc, labelnames=labelnames)

        self.gauge_scheduler_running = gauge("vllm:num_requests_running", "Number of requests currently running on GPU.")
        self.gauge_scheduler_swapped = gauge("vllm:num_requests_swapped", "Number of requests swapped to CPU.")
        self.gauge_scheduler_waiting = gauge("vllm:num_requests_waiting", "Number of requests waiting to be processed.")
        self.gauge_gpu_cache_usage = gauge("vllm:gpu_cache_usage_perc", "GPU KV-cache usage. 1 means 100 percent usage.")
        self.gauge_cpu_cache_usage = gauge("vllm:cpu_cache_usage_perc", "CPU KV-cache usage. 1 means 100 percent usage.")
        self.counter_prompt_tokens = Counter("vllm:prompt_tokens_total", "Number of prefill tokens processed.",
                                             labelnames=labelnames)
        self.counter_generation_tokens = Counter("vllm:generation_tokens_total",
                                                 "Number of generation tokens processed.", labelnames=labelnames)
        self.histogram_time_to_first_token = Histogram("vllm:time_to_first_token_seconds",
                                                       "Histogram of time to first token in seconds.",
                                                       labelnames=labelnames, buckets=_TTFT_BUCKETS)
        self.histogram_time_per_output_token = Histogram("vllm:time_per_output_token_seconds",
                                                         "Histogram of time per output token in seconds.",
                                                         labelnames=labelnames, buckets=_TPOT_BUCKETS)
        self.histogram_e2e_request_latency = Histogram("vllm:e2e_request_latency_seconds",
                                                       "Histogram of end to end request latency in seconds.",
                                                       labelnames=labelnames, buckets=_E2E_BUCKETS)
        self.gauge_avg_prompt_throughput = gauge("vllm:avg_prompt_throughput_toks_per_s",
                                                 "Average prefill throughput in tokens/s.")
        self.gauge_avg_generation_throughput = gauge("vllm:avg_generation_throughput_toks_per_s",
                                                     "Average generation throughput in tokens/s.")


_current: Optional["StatLogger"] = None


def ensure_registered(model_name: str) -> None:
    """Makes the series exist (at zero) even when no engine logs statistics yet."""
    if _PROM and _current is None:
        StatLogger(local_interval=float("inf"), labels={"model_name": model_name})


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
    def __init__(self, local_interval: float, labels: Optional[Dict[str, str]] = None) -> None:
        global _current
        self.local_interval = local_interval
        self.last_local_log = time.monotonic()
        self.num_prompt_tokens: List[int] = []
        self.num_generation_tokens: List[int] = []
        self.labels = dict(labels or {})
        self.metrics = Metrics(list(self.labels)) if _PROM else None
        if self.metrics is not None:
            m = self.metrics
            # touch every labelled child so that the exposition shows the series before the first event
            for c in (m.gauge_scheduler_running, m.gauge_scheduler_swapped, m.gauge_scheduler_waiting,
                      m.gauge_gpu_cache_usage, m.gauge_cpu_cache_usage, m.counter_prompt_tokens,
                      m.counter_generation_tokens, m.histogram_time_to_first_token, m.histogram_time_per_output_token,
                      m.histogram_e2e_request_latency, m.gauge_avg_prompt_throughput, m.gauge_avg_generation_throughput):
                self._of(c)
        _current = self

    def _of(self, collector):
        return collector.labels(**self.labels) if self.labels else collector

    def info(self, kind: str, obj) -> None:
        if kind == "cache_config" and self.metrics is not None:
            self.metrics.info_cache_config.info(obj.metrics_info())

    def _log_prometheus(self, stats: Stats) -> None:
        m = self.metrics
        self._of(m.gauge_scheduler_running).set(stats.num_running)
        self._of(m.gauge_scheduler_swapped).set(stats.num_swapped)
        self._of(m.gauge_scheduler_waiting).set(stats.num_waiting)
        self._of(m.gauge_gpu_cache_usage).set(stats.gpu_cache_usage)
        self._of(m.gauge_cpu_cache_usage).set(stats.cpu_cache_usage)
        self._of(m.counter_prompt_tokens).inc(stats.num_prompt_tokens)
        self._of(m.counter_generation_tokens).inc(stats.num_generation_tokens)
        for v in stats.time_to_first_tokens:
            self._of(m.histogram_time_to_first_token).observe(v)
        for v in stats.time_per_output_tokens:
            self._of(m.histogram_time_per_output_token).observe(v)
        for v in stats.time_e2e_requests:
            self._of(m.histogram_e2e_request_latency).observe(v)

    def log(self, stats: Stats) -> None:
        """Every step: Prometheus; every ``local_interval`` seconds of the engine's clock (``stats.now``):
        the interval's throughputs, to the legacy gauges and to stdout."""
        if self.metrics is not None:
            self._log_prometheus(stats)
        self.num_prompt_tokens.append(stats.num_prompt_tokens)
        self.num_generation_tokens.append(stats.num_generation_tokens)
        elapsed = stats.now - self.last_local_log
        if elapsed > self.local_interval:
            prompt_tput = sum(self.num_prompt_tokens) / elapsed
            gen_tput = sum(self.num_generation_tokens) / elapsed
            if self.metrics is not None:
                self._of(self.metrics.gauge_avg_prompt_throughput).set(prompt_tput)
                self._of(self.metrics.gauge_avg_generation_throughput).set(gen_tput)
            print(f"Avg prompt throughput: {prompt_tput:.1f} tokens/s, "
                  f"Avg generation throughput: {gen_tput:.1f} tokens/s, "
                  f"Running: {stats.num_running} reqs, Swapped: {stats.num_swapped} reqs, "
                  f"Pending: {stats.num_waiting} reqs, GPU KV cache usage: {stats.gpu_cache_usage * 100:.1f}%, "
                  f"CPU KV cache usage: {stats.cpu_cache_usage * 100:.1f}%", flush=True)
            self.num_prompt_tokens, self.num_generation_tokens = [], []
            self.last_local_log = stats.now
