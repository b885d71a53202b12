"""One tensor-parallel replica (co-located or split pools) with pipelined and with lock-step engine calls,
the host time of every engine call printed: tells a control-plane stall from a slow step.

  python tools/tp_pipeline_probe.py --kind colocated --tp 2 --model llama-2-7b [--one-gpu]
"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-splitwise_amd"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="colocated", choices=["colocated", "split"])
    ap.add_argument("--tp", type=int, default=2)
    ap.add_argument("--model", default="llama-2-7b")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--concurrency", type=int, default=16)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--one-gpu", action="store_true")
    args = ap.parse_args()
    if args.one_gpu:
        os.environ["SPLITWISE_BENCH_ONE_GPU"] = "1"
    os.environ["SPLITWISE_BENCH_VERBOSE"] = "1"
    from splitwise_amd.executor import bench_launch
    import torch.distributed as dist
    world = args.tp * (2 if args.kind == "split" else 1)
    devices = [0] * world if args.one_gpu else list(range(world))
    store = dist.TCPStore("127.0.0.1", 0, None, is_master=True, wait_for_workers=False)
    for pipelined in (True, False):
        leg = {"name": f"probe_{'pipelined' if pipelined else 'lockstep'}", "replicas": [
            bench_launch._replica(args.kind, devices, args.model, args.dtype, tp=args.tp)]}
        common = {"concurrency": args.concurrency, "prompt_len": 772, "decode_len": 170, "steps": args.steps,
                  "warmup": args.warmup, "max_num_batched_tokens": 4096, "prefill_batch_tokens": 2300,
                  "prefill_max_wait": 0.08, "pipeline_steps": pipelined, "enforce_eager": False, "no_fill": False,
                  "no_roofline": True}
        with tempfile.TemporaryDirectory() as work:
            res = bench_launch.run_leg(leg, common, store.port, work, 300.0, os.path.join(ROOT, "bench.py"), [])[0]
        keep = {k: res[k] for k in ("elapsed_s", "good_tokens", "total_tokens", "ttft_p50_ms", "tpot_p50_ms",
                                    "tpot_p99_ms", "prompt_steps", "decode_steps", "step_ms_max")}
        print(json.dumps({"leg": leg["name"], **keep}), flush=True)
        print("  step_ms:", " ".join(f"{v:.1f}" for v in res.get("step_ms", [])), flush=True)


if __name__ == "__main__":
    main()
