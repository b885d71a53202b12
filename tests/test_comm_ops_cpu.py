"""The tensor-parallel collectives and group layouts on CPU (gloo), after the reference's
tests/distributed/test_comm_ops.py:19-54 (all-reduce and all-gather workers; its third worker tests
broadcast_tensor_dict, which this build does not have: step commands travel over the executors' own channels).
World 4 as two co-located TP2 groups and as a TP2 prompt pool + a TP2 token pool; world 6 as one prompt pool feeding
two token pools."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank: int, world: int, tp: int, split: bool, token_pools: int, init: str, out_dir: str) -> None:
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import conftest  # noqa: F401
    from splitwise_amd.model_executor import parallel_state as ps
    dist.init_process_group("gloo", init_method=init, rank=rank, world_size=world)
    ps.initialize_model_parallel(tp, split, token_pools=token_pools)
    group_first = rank // tp * tp
    members = list(range(group_first, group_first + tp))
    assert ps.get_tensor_model_parallel_world_size() == tp and ps.get_tensor_model_parallel_rank() == rank - group_first

    # all-reduce: sums over the rank's own group only (test_comm_ops.py:19-31)
    x = torch.arange(8, dtype=torch.float32) * (rank + 1)
    y = ps.tensor_model_parallel_all_reduce(x.clone())
    assert torch.equal(y, torch.arange(8, dtype=torch.float32) * sum(m + 1 for m in members))

    # all-gather along every dimension (test_comm_ops.py:35-53)
    for dim in (0, 1, 2, -1):
        t = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(2, 3, 4) * (rank + 1)
        expected = torch.cat([torch.arange(24, dtype=torch.float32).reshape(2, 3, 4) * (m + 1) for m in members], dim=dim)
        assert torch.equal(ps.tensor_model_parallel_all_gather(t, dim), expected)
        got = ps.tensor_model_parallel_gather(t, dst=0, dim=dim)
        assert (got is None) == (rank != group_first)
        if got is not None:
            assert torch.equal(got, expected)

    # the first rank's tensor on every rank of the group
    z = torch.full((5,), float(rank))
    assert torch.equal(ps.tensor_model_parallel_broadcast_from_first(z), torch.full((5,), float(group_first)))

    # stage groups: the whole world when co-located; with split pools the prompt pool, then each token pool
    stage = ps.get_stage_parallel_group()
    pool = world // (1 + token_pools) if split else world
    ones = torch.ones(1)
    dist.all_reduce(ones, group=stage)
    assert int(ones) == pool
    first = torch.tensor([float(rank)])
    dist.all_reduce(first, op=dist.ReduceOp.MIN, group=stage)
    assert int(first) == rank // pool * pool
    ps.destroy_model_parallel()
    dist.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").close()


@pytest.mark.parametrize("world,tp,split,token_pools", [(4, 2, False, 1), (4, 2, True, 1), (6, 2, True, 2)])
def test_collectives_and_group_layouts(world, tp, split, token_pools):
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, tp, split, token_pools, "file://" + os.path.join(tmp, "rendezvous"), tmp),
                 nprocs=world, join=True)
        assert sorted(os.listdir(tmp)).count("rendezvous") <= 1
        assert all(os.path.exists(os.path.join(tmp, f"ok{r}")) for r in range(world))
