"""Dry runs of the multi-GPU benchmark launcher on CPU: ``bench.py --gpus N`` with stand-in workers
(tests/split_fakes.py) at N = 2, 4 and 8, started the two ways a driver may start it -- plainly, and under
``python -m torch.distributed.run``.  Everything but the workers is the product code: the leg plan, fresh
replica processes per leg, the TCP-store meeting points around the timed steps, executors that spawn their own
workers (pairs, one prompt worker feeding two token workers, tensor-parallel pools over gloo), result
aggregation into ONE JSON line, and the failure path (a leg that cannot start is reported in the line, the
others still run)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--worker-cls", "split_fakes:FakeWorker", "--steps", "12", "--warmup", "3", "--concurrency", "6",
         "--prompt-len", "40", "--decode-len", "10", "--max-num-batched-tokens", "2048"]


def _run(cmd, extra_env=None, timeout=420):
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tests"), SPLIT_FAKE_BLOCKS="256")
    for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(name, None)
    env.update(extra_env or {})
    proc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    return proc, lines


def test_leg_plans():
    import conftest  # noqa: F401
    from splitwise_amd.executor.bench_launch import build_plan
    plan = {leg["name"]: leg for leg in build_plan(8)}
    assert list(plan) == ["pairs", "colocated", "asymmetric", "colocated_overlap", "70b_tp4_pools", "70b_tp4_colocated"]
    assert all(r["overlap"] for r in plan["colocated_overlap"]["replicas"])
    assert plan["colocated_overlap"]["layout"] == "8x(co-located, prompt beside decode)"
    assert [r["devices"] for r in plan["pairs"]["replicas"]] == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert [r["devices"] for r in plan["asymmetric"]["replicas"]] == [[0, 1, 2], [3, 4, 5], [6, 7]]
    assert plan["asymmetric"]["layout"] == "2x(1P+2D) + 1P+1D"
    assert plan["70b_tp4_pools"]["replicas"][0]["tp"] == 4 and plan["70b_tp4_pools"]["layout"] == "TP4+TP4"
    # every GPU of the node is used by every leg, exactly once
    for n in (2, 3, 4, 8):
        for leg in build_plan(n):
            used = sorted(d for r in leg["replicas"] for d in r["devices"])
            assert used == list(range(n)), (n, leg["name"], used)
    four = {leg["name"]: leg for leg in build_plan(4)}
    assert four["13b_gptq_fp8_tp2_pools"]["replicas"][0]["quantization"] == "gptq"
    assert four["13b_gptq_fp8_tp2_pools"]["replicas"][0]["kv_cache_dtype"] == "fp8_e5m2"
    assert [leg["name"] for leg in build_plan(4, legs="primary")] == ["pairs", "colocated"]
    with pytest.raises(SystemExit):
        build_plan(2, legs="no_such_leg")


@pytest.mark.parametrize("n", [2, 4, 8])
def test_plain_python_launch_prints_one_line_with_every_leg(n):
    proc, lines = _run([sys.executable, "bench.py", "--gpus", str(n)] + SMALL)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert len(lines) == 1, lines                                   # stdout carries the JSON line alone
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["value"] > 0 and line["scaling"] == "weak"
    assert line["metric"].startswith("goodput") and line["unit"] == "tok/s"
    legs = line["legs"]
    expected = {2: {"pairs", "colocated", "colocated_overlap"},
                4: {"pairs", "colocated", "asymmetric", "colocated_overlap", "13b_gptq_fp8_tp2_pools",
                    "13b_gptq_fp8_pairs", "13b_gptq_fp8_colocated"},
                8: {"pairs", "colocated", "asymmetric", "colocated_overlap", "70b_tp4_pools", "70b_tp4_colocated"}}[n]
    assert set(legs) == expected
    for name, leg in legs.items():
        assert "error" not in leg and "skipped" not in leg, (name, leg)
        assert leg["n_gpus"] == n and leg["value"] > 0 and leg["steady_state"], (name, leg)
        if "colocated" not in name:
            assert leg["vs_colocated"] > 0 and leg["kv_bytes_pushed"] > 0, (name, leg)
        if name == "colocated_overlap":
            assert leg["vs_colocated"] > 0 and leg["kv_bytes_pushed"] == 0 and leg["pipelined"], leg
    assert line["vs_colocated"] == legs["pairs"]["vs_colocated"]
    assert line["config"]["parallelism"] == {2: "1P+1D", 4: "2x(1P+1D)", 8: "4x(1P+1D)"}[n]


def test_launch_under_torchrun_gives_the_same_single_line():
    proc, lines = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2",
                        "--legs", "primary"] + SMALL)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and set(line["legs"]) == {"pairs", "colocated"}


def test_a_leg_that_cannot_start_is_reported_and_the_run_goes_on():
    # the stand-in worker of rank 2 dies during the KV-transport handshake: only the trio of the asymmetric leg
    # has a rank 2, so that leg fails at set-up; the line still carries the headline leg and says what went wrong
    proc, lines = _run([sys.executable, "bench.py", "--gpus", "3", "--legs", "pairs,asymmetric"] + SMALL,
                       extra_env={"SPLIT_FAKE_DIE_IN_HANDSHAKE": "2"})
    assert len(lines) == 1, (lines, proc.stderr[-2000:])
    line = json.loads(lines[0])
    assert proc.returncode == 0 and line["value"] > 0
    failed = line["legs"]["asymmetric"]
    assert "value" not in failed and "token worker 1" in failed["error"], failed


def test_headline_falls_back_to_the_runtime_transport_and_then_to_the_colocated_leg():
    # the token worker dies in the handshake of the kernel transport only: the pairs run again over the runtime
    # transport and that leg is the headline, named as such
    proc, lines = _run([sys.executable, "bench.py", "--gpus", "2"] + SMALL,
                       extra_env={"SPLIT_FAKE_DIE_WITH_TRANSPORT": "kernel"})
    assert len(lines) == 1, (lines, proc.stderr[-2000:])
    line = json.loads(lines[0])
    assert proc.returncode == 0 and line["value"] > 0
    assert list(line["legs"]) == ["pairs", "pairs_runtime_transport", "colocated", "colocated_overlap"]
    assert "error" in line["legs"]["pairs"] and line["legs"]["pairs_runtime_transport"]["value"] == line["value"]
    assert line["config"]["headline_leg"] == "pairs_runtime_transport" and line["config"]["kv_transport"] == "runtime"
    assert "pairs) failed" in line["error"] and line["vs_colocated"] > 0
    # it dies whatever the transport: the co-located leg of the same GPUs is what the line reports, and says so
    proc, lines = _run([sys.executable, "bench.py", "--gpus", "2"] + SMALL,
                       extra_env={"SPLIT_FAKE_DIE_IN_HANDSHAKE": "1"})
    assert len(lines) == 1, (lines, proc.stderr[-2000:])
    line = json.loads(lines[0])
    assert proc.returncode == 0 and line["value"] == line["legs"]["colocated"]["value"] > 0
    assert line["config"]["headline_leg"] == "colocated" and line["config"]["parallelism"] == "2x(co-located)"
    assert "no KV migration" in line["config"]["workload"] and "failed" in line["error"]
    assert line["kv_migration"] is None and line["vs_colocated"] is None
    # ... even when the failed legs have used the run's time budget up: the line is owed a value
    proc, lines = _run([sys.executable, "bench.py", "--gpus", "2", "--budget-s", "1"] + SMALL,
                       extra_env={"SPLIT_FAKE_DIE_IN_HANDSHAKE": "1"})
    line = json.loads(lines[0])
    assert proc.returncode == 0 and line["value"] == line["legs"]["colocated"]["value"] > 0
    assert "skipped" in line["legs"]["pairs_runtime_transport"] and line["config"]["headline_leg"] == "colocated"
