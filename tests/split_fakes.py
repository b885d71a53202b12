"""CPU stand-ins for the workers of a split deployment, used to drive the REAL control plane
(engine -> SplitwiseExecutor / PoolExecutor -> channel -> worker_loop) across processes without a GPU.
With tensor-parallel pools (PoolExecutor) rank r of the prompt pool pushes to rank r + tp, every
rank joins the real process groups over gloo, and every step all-reduces the sampled token over the
pool's tensor-parallel group: ranks that disagree (a push that went to the wrong peer, a group that
spans the wrong ranks) fail the step.

The "KV cache" of a worker is a file-backed int64 array [num_blocks, block_size] holding token
ids, plus an epoch flag per hand-off slot; the prompt side maps the token side's file (the
stand-in for the IPC mapping) and copies block runs into it.  The token side computes every new
token FROM ITS CACHE, so a missing or late block push, a wrong block table, run or epoch shows
up as a wrong token or a failed wait.
"""
import os
import time
from typing import Dict, List, Optional

import numpy as np

from splitwise_amd.sequence import Logprob, SamplerOutput, SequenceGroupOutput, SequenceOutput

VOCAB = 512


def next_token(tokens: List[int]) -> int:
    acc = 0
    for i, t in enumerate(tokens):
        acc = (acc * 31 + t + i) % 1000003
    return acc % VOCAB


class _Deferred:
    """Stands for samples that are still on the device."""

    def __init__(self, out: SamplerOutput) -> None:
        self.out = out

    def done(self) -> bool:
        return True

    def wait(self) -> None:
        pass

    def resolve(self) -> SamplerOutput:
        return self.out


class _FakeComm:
    def __init__(self) -> None:
        self.bytes_pushed = 0


class _FakeCommManager:
    def __init__(self) -> None:
        self.kvcache_comm = _FakeComm()

    def check_status(self) -> None:
        pass


class _FakeRunner:
    enforce_eager = True


class FakeWorker:
    def __init__(self, model_config, parallel_config, scheduler_config, device_config, local_rank=0, rank=0,
                 kv_cache_dtype="auto", is_driver_worker=True, kv_transport_init=None, distributed_init_method=None,
                 distributed_store=None) -> None:
        self.rank = rank
        self.parallel_config = parallel_config
        self.tp = parallel_config.tensor_parallel_size
        self.world = parallel_config.world_size            # tp * (1 + token pools)
        self.mixed = not parallel_config.sep_prompt_token      # co-located: one pool does both kinds of step
        self.is_prompt = rank < self.tp and not self.mixed
        # a prompt rank is paired with the same rank of every token pool, a token rank with its prompt rank
        self.peers = list(range(rank + self.tp, self.world, self.tp)) if self.is_prompt else [rank % self.tp]
        self.peer = self.peers[0]
        self.staged = bool(getattr(parallel_config, "staged_prompt_cache", False))
        self.distributed_init_method = distributed_init_method
        self.distributed_store = distributed_store
        self.exchange = kv_transport_init["exchange"] if kv_transport_init else None
        self.transport = kv_transport_init.get("transport", "kernel") if kv_transport_init else None
        self.dir = os.environ.get("SPLITWISE_DRYRUN_DIR") or os.environ["SPLIT_FAKE_DIR"]
        self.model_runner = _FakeRunner()
        self.kvcache_comm_manager = _FakeCommManager() if not self.mixed else None
        self.cache = self.flags = None
        self.cached_prompt_tokens = 0          # prompt tokens taken from blocks reported as computed
        self.peer_caches, self.peer_flag_arrays = [], []
        self.block_size = 16
        self.log: List[str] = []

    # ---- life cycle, as Worker
    def init_model(self) -> None:
        if self.tp == 1:
            return
        # the product's own group set-up (worker.py:66-100 of this package), on gloo
        import datetime

        import torch.distributed as dist

        from splitwise_amd.model_executor import parallel_state
        timeout = datetime.timedelta(seconds=120)
        if self.distributed_store is not None:
            dist.init_process_group("gloo", world_size=self.world, rank=self.rank, timeout=timeout,
                                    store=dist.PrefixStore("default_pg", self.distributed_store))
        else:
            dist.init_process_group("gloo", world_size=self.world, rank=self.rank, timeout=timeout,
                                    init_method=self.distributed_init_method)
        parallel_state.initialize_model_parallel(self.tp, not self.mixed,
                                                 token_pools=max(self.parallel_config.num_token_pools, 1))
        assert parallel_state.get_tensor_model_parallel_rank() == self.rank % self.tp

    def _agree(self, token: int) -> int:
        """All ranks of the pool must have computed the same token: SUM over the tensor-parallel group."""
        if self.tp == 1:
            return token
        import torch

        from splitwise_amd.model_executor import parallel_state
        total = parallel_state.tensor_model_parallel_all_reduce(torch.tensor([token, 1], dtype=torch.int64))
        assert int(total[1]) == self.tp, f"the all-reduce spanned {int(total[1])} ranks, not {self.tp}"
        if int(total[0]) != token * self.tp:
            raise RuntimeError(f"rank {self.rank} sampled {token}, its pool's sum is {int(total[0])}")
        return token

    def load_model(self) -> None:
        pass

    def profile_num_available_blocks(self, block_size, gpu_memory_utilization, cpu_swap_space, cache_dtype):
        small = int(os.environ.get("SPLIT_FAKE_BLOCKS", "64"))
        return (small if self.is_prompt else small + 16), 8  # the executor must take the smaller one

    def get_cache_block_size_bytes(self, block_size, cache_dtype) -> int:
        return block_size * 8

    def init_cache_engine(self, cache_config) -> None:
        self.block_size = cache_config.block_size
        n = cache_config.num_gpu_blocks
        self.cache = np.lib.format.open_memmap(os.path.join(self.dir, f"cache{self.rank}.npy"), mode="w+",
                                               dtype=np.int64, shape=(n, self.block_size))
        self.cache[:] = -1
        self.flags = np.lib.format.open_memmap(os.path.join(self.dir, f"flags{self.rank}.npy"), mode="w+",
                                               dtype=np.int64, shape=(256,))
        self.flags[:] = 0

    def warm_up_model(self) -> None:
        pass

    def setup_kvcache_comm(self) -> None:
        if self.mixed:
            return
        if os.environ.get("SPLIT_FAKE_DIE_IN_HANDSHAKE") == str(self.rank):
            os._exit(3)                                    # a worker that dies before it has handed anything in
        if os.environ.get("SPLIT_FAKE_DIE_WITH_TRANSPORT") == self.transport and not self.is_prompt:
            os._exit(3)                                    # a transport that does not come up on this node
        def swap(obj, index):
            return self.exchange(obj) if len(self.peers) == 1 else self.exchange(obj, index)

        for index, peer_rank in enumerate(self.peers):
            peer = swap({"rank": self.rank, "blocks": int(self.cache.shape[0])}, index)
            assert peer["rank"] == peer_rank, peer
            # without staging a block number means the same block on both sides: equal sizes
            assert self.staged or peer["blocks"] == self.cache.shape[0], peer
            if self.is_prompt:
                self.peer_caches.append(np.load(os.path.join(self.dir, f"cache{peer_rank}.npy"), mmap_mode="r+"))
                self.peer_flag_arrays.append(np.load(os.path.join(self.dir, f"flags{peer_rank}.npy"), mmap_mode="r+"))
        for index in range(len(self.peers)):
            swap({"rank": self.rank, "ready": True}, index)

    def destroy_kvcache_comm(self) -> None:
        pass

    # ---- one step
    def _write(self, block_table: List[int], pos: int, token: int) -> None:
        self.cache[block_table[pos // self.block_size], pos % self.block_size] = token

    def _read(self, block_table: List[int], n: int) -> List[int]:
        return [int(self.cache[block_table[i // self.block_size], i % self.block_size]) for i in range(n)]

    def execute_model(self, seq_group_metadata_list=None, blocks_to_swap_in=None, blocks_to_swap_out=None,
                      blocks_to_copy=None, blocks_to_nw=None, nw_epochs=None, defer=False, nw_peer=0,
                      track_pending=True):
        out = self._execute(seq_group_metadata_list, blocks_to_nw, nw_epochs, nw_peer)
        # like the device, this worker remembers the tokens it sampled last: a deferred step's tokens
        # reach the engine one call later, and the next command names them by a placeholder (-1)
        # (track_pending off: a prompt step that runs beside the decode steps of the same worker)
        if track_pending:
            self._last_samples = {g.samples[0].parent_seq_id: g.samples[0].output_token for g in out}
        return _Deferred(out) if defer else out

    def _execute(self, seq_group_metadata_list, blocks_to_nw, nw_epochs, nw_peer=0) -> SamplerOutput:
        blocks_to_nw = blocks_to_nw or {}
        outputs = []
        if self.is_prompt or (self.mixed and seq_group_metadata_list[0].is_prompt):
            for meta in seq_group_metadata_list:
                assert meta.is_prompt
                (seq_id, data), = meta.seq_data.items()
                tokens = data.get_token_ids()
                table = meta.block_tables[seq_id]
                # blocks reported as computed are not written again: their content must be in the cache already,
                # and the sampled token is then taken from what the cache holds
                skip = len(getattr(meta, "computed_block_nums", None) or []) * self.block_size
                self.cached_prompt_tokens += skip
                for pos, tok in enumerate(tokens):
                    if pos >= skip:
                        self._write(table, pos, tok)
                tok = self._agree(next_token(self._read(table, len(tokens)) if skip else tokens))
                outputs.append(SequenceGroupOutput([SequenceOutput(seq_id, tok, {tok: Logprob(0.0)})]))
            if self.mixed:
                return SamplerOutput(outputs)
            time.sleep(0.002)                              # the decode step on the other side overlaps this one
            peer_cache, peer_flags = self.peer_caches[nw_peer], self.peer_flag_arrays[nw_peer]
            for slot, runs in blocks_to_nw.items():
                for run in runs:                           # (start, count), or (source start, destination start, count)
                    src, dst, count = (run[0], run[0], run[1]) if len(run) == 2 else run
                    peer_cache[dst:dst + count] = self.cache[src:src + count]
                    self.kvcache_comm_manager.kvcache_comm.bytes_pushed += count * self.block_size * 8
                peer_cache.flush()
                peer_flags[slot] = nw_epochs[slot]         # signal after the data
                peer_flags.flush()
            if self.staged:
                self.cache[:] = -7                         # a staged prompt cache keeps nothing between steps
            return SamplerOutput(outputs)
        # token side: wait for every hand-off this step depends on, then decode from the cache
        for slot in blocks_to_nw:
            deadline = time.monotonic() + 20.0
            while int(self.flags[slot]) < nw_epochs[slot]:
                if time.monotonic() > deadline:
                    raise RuntimeError(f"hand-off wait timed out on slot {slot} (epoch {nw_epochs[slot]})")
                time.sleep(0.0005)
        for meta in seq_group_metadata_list:
            assert not meta.is_prompt
            (seq_id, data), = meta.seq_data.items()
            # a decode command carries the sequence's length and last token only: the history is what
            # the cache holds (pushed by the prompt side, then written step by step here), so a lost or
            # misplaced block shows up as a wrong token
            n, last = data.get_len(), data.get_last_token_id()
            if last == -1:
                last = self._last_samples[seq_id]
            table = meta.block_tables[seq_id]
            self._write(table, n - 1, last)                          # this step's own KV entry
            cached = self._read(table, n)
            tok = self._agree(next_token(cached))
            outputs.append(SequenceGroupOutput([SequenceOutput(seq_id, tok, {tok: Logprob(0.0)})]))
        return SamplerOutput(outputs)
