"""GPU parity: all-reduce by direct peer reads (the C-ABI's sw_custom_all_reduce) between 2 and 4
processes, bit for bit against the CPU restatement of the reference's reduction order
(csrc/custom_all_reduce.cuh:204-290), as tests/distributed/test_custom_all_reduce.py:24-57 checks
it against the library all-reduce.  All ranks share the one GPU of the test box: the peers'
buffers are mapped through the same IPC calls as across GPUs."""
import os
import tempfile

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 4])
def test_all_reduce_matches_reference_order_bit_for_bit(world):
    import ar_worker
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    with tempfile.TemporaryDirectory() as tmp:
        # rendezvous through a file: no port to pick before the ranks exist
        init_method = "file://" + os.path.join(tmp, "rendezvous")
        procs = [ctx.Process(target=ar_worker.run, args=(r, world, init_method, queue), daemon=True)
                 for r in range(world)]
        for p in procs:
            p.start()
        try:
            results = [queue.get(timeout=150) for _ in range(world)]
        finally:
            for p in procs:
                p.join(timeout=10)
                if p.is_alive():
                    p.kill()
    for rank, bad in sorted(results):
        assert bad == [], f"rank {rank}: {bad}"


@pytest.mark.parametrize("sabotaged_rank", [-1, 1])
def test_the_path_is_only_used_after_a_known_answer_check(sabotaged_rank):
    """init_custom_ar reduces a known tensor in both forms before the path is trusted.  With every rank honest
    the path comes up; with one rank whose peer-read sums are wrong (as on a node whose links do not behave) NO
    rank uses it, and the tensor-parallel all-reduce still gives the right sums through the library."""
    import ar_worker
    world = 2
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    with tempfile.TemporaryDirectory() as tmp:
        init_method = "file://" + os.path.join(tmp, "rendezvous")
        procs = [ctx.Process(target=ar_worker.run_init, args=(r, world, init_method, queue, sabotaged_rank),
                             daemon=True) for r in range(world)]
        for p in procs:
            p.start()
        try:
            results = dict(queue.get(timeout=150) for _ in range(world))
        finally:
            for p in procs:
                p.join(timeout=10)
                if p.is_alive():
                    p.kill()
    for rank in range(world):
        assert results[rank] == (sabotaged_rank < 0, True), (rank, results[rank])


def _rccl_single_rank(init_method: str, queue) -> None:
    """As much of the default ("nccl" = RCCL) collective path as one GPU allows: backend start-up, the
    tensor-parallel group helpers, a message above the peer-read kernel's 8 MiB limit (the size class that
    always goes to RCCL), all-gather and broadcast on a sub-group.  RCCL refuses two ranks on one
    device, so the group has one member; what this guards is that the first multi-GPU run is not also
    the first time this code and the library's communicator set-up execute."""
    try:
        import os
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, here)
        import conftest  # noqa: F401
        import torch
        import torch.distributed as dist
        from splitwise_amd.model_executor import parallel_state
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, init_method=init_method)
        assert dist.get_backend() == "nccl"
        group = dist.new_group([0])
        big = torch.full((9 * 1024 * 1024 // 2,), 1.5, dtype=torch.bfloat16, device="cuda")      # 9 MiB
        dist.all_reduce(big, group=group)
        parts = [torch.empty(8, device="cuda")]
        dist.all_gather(parts, torch.arange(8, dtype=torch.float32, device="cuda"), group=group)
        dist.broadcast(big, src=0, group=group)
        torch.cuda.synchronize()
        ok = bool((big == 1.5).all()) and parts[0].tolist() == list(range(8))
        # the package's own entry points with a group of one are pass-throughs
        parallel_state.initialize_model_parallel(1)
        x = torch.ones(4, device="cuda")
        ok = ok and parallel_state.tensor_model_parallel_all_reduce(x) is x
        ok = ok and parallel_state.tensor_model_parallel_all_gather(x) is x
        dist.destroy_process_group()
        queue.put("ok" if ok else "wrong values")
    except BaseException:                         # noqa: BLE001
        import traceback
        queue.put(traceback.format_exc())


def test_rccl_backend_starts_and_reduces_on_one_gpu():
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    with tempfile.TemporaryDirectory() as tmp:
        proc = ctx.Process(target=_rccl_single_rank, args=("file://" + os.path.join(tmp, "rendezvous"), queue), daemon=True)
        proc.start()
        try:
            verdict = queue.get(timeout=100)
        finally:
            proc.join(timeout=10)
            if proc.is_alive():
                proc.kill()
    assert verdict == "ok", verdict
