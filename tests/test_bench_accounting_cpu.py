"""The closed-loop driver of bench.py against a stand-in engine with pipelined steps: the ramp admits at the
steady state's rate, and a timed window is credited with exactly the tokens of the steps launched inside it."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conftest  # noqa: F401,E402
import bench  # noqa: E402


class _Group:
    def __init__(self, request_id, max_tokens):
        self.request_id = request_id
        self.max_tokens = max_tokens
        self.tokens = []
        self.launched = 0            # tokens of launched steps, collected or not


class PipelinedEngine:
    """step() launches a step over every live request and returns the outputs of the step launched by the
    previous call (one token per request of that step), like the engine with pipeline_steps."""

    def __init__(self):
        self.model_config = types.SimpleNamespace(get_vocab_size=lambda: 100)
        self.scheduler = types.SimpleNamespace(waiting=[])
        self.groups = {}
        self._inflight = None
        self.last_scheduled_prompt_run = False
        self.launch_log = []         # number of sequences of every launched step

    def add_request(self, request_id, prompt, params, prompt_token_ids=None):
        self.groups[request_id] = _Group(request_id, params.max_tokens)

    def inflight_request_ids(self):
        return [g.request_id for g, _ in self._inflight[2]] if self._inflight is not None else []

    def step(self):
        previous = self._inflight
        live = [g for g in self.groups.values() if g.launched < g.max_tokens]
        for g in live:
            g.launched += 1
        self._inflight = (None, None, [(g, None) for g in live]) if live else None
        self.launch_log.append(len(live))
        outs = []
        if previous is not None:
            for g, _ in previous[2]:
                g.tokens.append(1)
                finished = len(g.tokens) == g.max_tokens
                outs.append(types.SimpleNamespace(request_id=g.request_id, finished=finished,
                                                  outputs=[types.SimpleNamespace(token_ids=list(g.tokens))]))
                if finished:
                    del self.groups[g.request_id]
        return outs


def test_ramp_reaches_the_target_and_windows_count_the_steps_launched_in_them():
    engine = PipelinedEngine()
    loop = bench.ClosedLoop(engine, concurrency=8, prompt_len=4, decode_len=16, seed=0)
    assert loop.fill()
    assert loop.ramp == 8 and loop.live >= 7          # (a request that finished in the last step is replaced by the next)
    # the ramp admits C / L = 0.5 requests per decode step: 8 requests need about 16 steps, not 8
    first_full = next(i for i, n in enumerate(engine.launch_log) if n == 8)
    assert 12 <= first_full <= 20
    for steps in (1, 2, 7, 20):
        for r in loop.req.values():
            r["window"] = 0
        before = len(engine.launch_log)
        loop.open_window()
        returned = sum(loop.step(True) for _ in range(steps))
        in_flight = loop.close_window()
        launched = sum(engine.launch_log[before:])
        assert returned + in_flight == launched
        assert sum(r["window"] for r in loop.req.values()) == launched
