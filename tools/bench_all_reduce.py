#!/usr/bin/env python3
"""Latency of the peer-read all-reduce between WORLD processes that share one GPU (what a
one-GPU box can measure: launch + barrier + local-memory traffic, no xGMI hop).
  python tools/bench_all_reduce.py [world]"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-splitwise_amd")]


def run(rank, world, port, queue):
    import torch
    import torch.distributed as dist
    from splitwise_amd.model_executor.custom_all_reduce import CustomAllreduce
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
    ar = CustomAllreduce(rank, world)
    rows = []
    for nbytes in (16 << 10, 64 << 10, 256 << 10, 1 << 20, 4 << 20, 8 << 20):
        x = torch.randn(nbytes // 2, device="cuda").to(torch.bfloat16)
        out = torch.empty_like(x)
        for algo in (1, 2) if world > 2 else (1,):
            for _ in range(5):
                ar.all_reduce_unreg(x, out, algorithm=algo)
            torch.cuda.synchronize()
            dist.barrier()
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 50
            start.record()
            for _ in range(iters):
                ar.all_reduce_unreg(x, out, algorithm=algo)
            end.record()
            end.synchronize()
            rows.append((nbytes, algo, start.elapsed_time(end) * 1e3 / iters))
    ar.check_status()
    dist.barrier()
    ar.close()
    if rank == 0:
        queue.put(rows)


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=run, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    rows = q.get(timeout=200)
    for p in procs:
        p.join(timeout=20)
    for nbytes, algo, us in rows:
        print(f"all_reduce world={world} bf16 {nbytes >> 10:6d} KiB  {'one stage ' if algo == 1 else 'two stages'}  "
              f"{us:8.1f} us per call (staging copy + kernel)")
