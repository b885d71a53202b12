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
