"""One rank of the custom all-reduce test: every rank can regenerate every rank's input from the
seeds, so each checks its own output against the CPU restatement, bit for bit."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "vllm-splitwise_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

CASES = [  # (number of elements, dtype name, algorithm: 0 = size rule, 1 = one stage, 2 = two stages)
    (8, "float16", 1), (4096, "float16", 1), (4096 + 8, "bfloat16", 1), (1000 * 8, "float32", 1),
    (64, "bfloat16", 2), (8 * 1021, "float16", 2), (5120 * 128, "bfloat16", 2), (4 * 333, "float32", 2),
    (5120 * 64, "float16", 0), (8192 * 512, "bfloat16", 0),
]


def _input(rank: int, case: int, numel: int, dtype):
    import torch
    g = torch.Generator().manual_seed(1000 * case + rank)
    return (torch.randn(numel, generator=g) * 3).to(dtype)


def run(rank: int, world: int, init_method: str, queue) -> None:
    try:
        import torch
        import torch.distributed as dist
        from oracle import all_reduce as oracle
        from splitwise_amd.model_executor.custom_all_reduce import CustomAllreduce
        torch.cuda.set_device(0)                  # every rank on the one GPU of the test box
        dist.init_process_group("gloo", rank=rank, world_size=world, init_method=init_method)
        ar = CustomAllreduce(rank, world, max_size=8192 * 1024)
        bad = []
        for rounds in range(2):                   # twice: the counters of a second pass start where the first ended
            for ci, (numel, dname, algo) in enumerate(CASES):
                dtype = getattr(torch, dname)
                inputs = [_input(r, ci, numel, dtype) for r in range(world)]
                x = inputs[rank].cuda()
                assert ar.should_custom_ar(x)
                out = ar.all_reduce_unreg(x, algorithm=algo).cpu()
                nbytes = numel * x.element_size()
                two = algo == 2 or (algo == 0 and world > 2 and nbytes >= (512 if world <= 4 else 256) * 1024)
                if numel * x.element_size() // 16 < world:
                    two = False
                want = oracle.two_stage(inputs) if two else oracle.one_stage(inputs)
                if not torch.equal(out.view(torch.uint8), want.view(torch.uint8)):
                    bad.append((rounds, ci, int((out != want).sum())))
        # the registered-buffer form: the producer writes into the staging area itself
        numel, dtype = 2048, torch.float16
        inputs = [_input(r, 99, numel, dtype) for r in range(world)]
        stage = ar.staging(inputs[rank])
        stage.copy_(inputs[rank].cuda())
        out = ar.all_reduce_reg(stage).cpu()
        if not torch.equal(out, oracle.one_stage(inputs)):
            bad.append(("reg", 0, 0))
        assert not ar.should_custom_ar(torch.zeros(7, dtype=torch.float16, device="cuda"))       # not 16-byte granular
        assert not ar.should_custom_ar(torch.zeros(8192 * 1024, dtype=torch.float32, device="cuda"))  # too large
        torch.cuda.synchronize()
        ar.check_status()
        dist.barrier()
        ar.close()
        dist.destroy_process_group()
        queue.put((rank, bad))
    except BaseException:                         # noqa: BLE001 - reported to the parent
        queue.put((rank, traceback.format_exc()))


def run_init(rank: int, world: int, init_method: str, queue, sabotaged_rank: int) -> None:
    """The way a worker brings the path up (init_custom_ar, behind its known-answer self-check), with the
    all-reduce of ``sabotaged_rank`` made to return wrong sums (-1: nobody's)."""
    try:
        import torch
        import torch.distributed as dist
        from splitwise_amd.model_executor import custom_all_reduce as car
        from splitwise_amd.model_executor import parallel_state
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world, init_method=init_method)
        parallel_state.initialize_model_parallel(world)
        if rank == sabotaged_rank:
            real = car.CustomAllreduce.all_reduce_unreg

            def wrong(self, inp, out=None, algorithm=0):
                return real(self, inp, out, algorithm) + 1

            car.CustomAllreduce.all_reduce_unreg = wrong
        car.init_custom_ar(rank, world, parallel_state.get_tensor_model_parallel_group())
        x = (torch.arange(4096, device="cuda") % 5 + rank).to(torch.float16)
        y = parallel_state.tensor_model_parallel_all_reduce(x.clone())
        want = ((torch.arange(4096, device="cuda") % 5) * world + world * (world - 1) // 2).to(torch.float16)
        sums_right = bool(torch.equal(y, want)) or rank == sabotaged_rank and car.is_initialized()
        queue.put((rank, (car.is_initialized(), sums_right)))
        dist.barrier()
        parallel_state.destroy_model_parallel()
        dist.destroy_process_group()
    except BaseException:                         # noqa: BLE001 - reported to the parent
        queue.put((rank, traceback.format_exc()))
