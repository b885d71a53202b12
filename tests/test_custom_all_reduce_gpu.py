"""GPU parity: all-reduce by direct peer reads (the C-ABI's sw_custom_all_reduce) between 2 and 4
processes, bit for bit against the CPU restatement of the reference's reduction order
(csrc/custom_all_reduce.cuh:204-290), as tests/distributed/test_custom_all_reduce.py:24-57 checks
it against the library all-reduce.  All ranks share the one GPU of the test box: the peers'
buffers are mapped through the same IPC calls as across GPUs."""
import socket

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_all_reduce_matches_reference_order_bit_for_bit(world):
    import ar_worker
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=ar_worker.run, args=(r, world, port, queue), daemon=True) for r in range(world)]
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
