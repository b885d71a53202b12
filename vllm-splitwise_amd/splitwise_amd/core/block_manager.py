"""Paged KV-cache block accounting.

Behaviour restated from vllm/core/block_manager.py:13-487 with prefix caching off
(the allocator is then its FIFO variant, :35-38; evictor vllm/core/evictor.py:122-152):

  * physical block numbers are handed out 0, 1, 2, ... until every block of the pool
    has been used once; only then are freed blocks reused, oldest-freed first;
  * a prompt needs one block per started block_size tokens; an append needs a new
    block when the sequence grows into a new logical block;
  * a block shared by forked sequences is copied on the first append (copy-on-write);
  * watermark: a prompt is admitted only if 1% of the pool stays free afterwards.

Blocks are plain integers with a reference count per pool; there are no per-block
objects.  When a sequence is freed its blocks return to the pool in block-table order
(the reference iterates a set of objects there, so its order depends on memory
addresses and cannot be reproduced; block numbering is identical to the reference's
until the first reuse of a freed block).
"""
import enum
from collections import deque
from typing import Deque, Dict, List, Optional, Tuple

from ..sequence import Sequence, SequenceGroup, SequenceStatus
from ..utils import Device


class BlockPool:
    """Free-block accounting of one device."""

    def __init__(self, device: Device, block_size: int, num_blocks: int) -> None:
        self.device = device
        self.block_size = block_size
        self.num_blocks = num_blocks
        self._next_fresh = 0
        self._freed: Deque[int] = deque()
        self._ref = [0] * num_blocks

    def allocate(self) -> int:
        if self._next_fresh < self.num_blocks:
            block = self._next_fresh
            self._next_fresh += 1
        elif self._freed:
            block = self._freed.popleft()
        else:
            raise ValueError("No usable cache memory left")
        self._ref[block] = 1
        return block

    def share(self, block: int) -> None:
        self._ref[block] += 1

    def ref_count(self, block: int) -> int:
        return self._ref[block]

    def free(self, block: int) -> None:
        if self._ref[block] == 0:
            raise ValueError(f"Double free! block {block} on {self.device.name} is already freed.")
        self._ref[block] -= 1
        if self._ref[block] == 0:
            self._freed.append(block)

    def get_num_free_blocks(self) -> int:
        return self.num_blocks - self._next_fresh + len(self._freed)


class AllocStatus(enum.Enum):
    """vllm/core/block_manager.py:116-127."""
    OK = enum.auto()
    LATER = enum.auto()
    NEVER = enum.auto()


class BlockSpaceManager:
    def __init__(self, block_size: int, num_gpu_blocks: int, num_cpu_blocks: int, watermark: float = 0.01,
                 sliding_window: Optional[int] = None, enable_caching: bool = False) -> None:
        if enable_caching:
            raise NotImplementedError("prefix caching is not built yet")
        assert watermark >= 0.0
        self.block_size = block_size
        self.num_total_gpu_blocks = num_gpu_blocks
        self.num_total_cpu_blocks = num_cpu_blocks
        self.block_sliding_window = None
        if sliding_window is not None:
            assert sliding_window % block_size == 0, (sliding_window, block_size)
            self.block_sliding_window = sliding_window // block_size
        self.watermark = watermark
        self.watermark_blocks = int(watermark * num_gpu_blocks)
        self.enable_caching = False
        self.gpu_allocator = BlockPool(Device.GPU, block_size, num_gpu_blocks)
        self.cpu_allocator = BlockPool(Device.CPU, block_size, num_cpu_blocks)
        self.block_tables: Dict[int, List[int]] = {}
        self._on_cpu: set = set()            # seq ids whose table holds CPU block numbers

    # ------------------------------------------------------------------ prompts
    def can_allocate(self, seq_group: SequenceGroup) -> AllocStatus:
        seq = seq_group.get_seqs(status=SequenceStatus.WAITING)[0]
        need = seq.num_logical_blocks
        if self.block_sliding_window is not None:
            need = min(need, self.block_sliding_window)
        if self.num_total_gpu_blocks - need < self.watermark_blocks:
            return AllocStatus.NEVER
        if self.gpu_allocator.get_num_free_blocks() - need >= self.watermark_blocks:
            return AllocStatus.OK
        return AllocStatus.LATER

    def allocate(self, seq_group: SequenceGroup) -> None:
        waiting = seq_group.get_seqs(status=SequenceStatus.WAITING)
        table: List[int] = []
        window = self.block_sliding_window
        for logical in range(waiting[0].num_logical_blocks):
            if window is not None and logical >= window:
                table.append(table[logical % window])
            else:
                table.append(self.gpu_allocator.allocate())
        # every waiting sequence of the group starts from the same prompt blocks; the
        # reference hands each a copy of the table without raising reference counts
        for seq in waiting:
            self.block_tables[seq.seq_id] = list(table)

    # ------------------------------------------------------------------ appends
    def can_append_slot(self, seq_group: SequenceGroup) -> bool:
        return seq_group.num_seqs(status=SequenceStatus.RUNNING) <= self.gpu_allocator.get_num_free_blocks()

    def append_slot(self, seq: Sequence) -> Optional[Tuple[int, int]]:
        """Makes room for the sequence's newest token; returns (src, dst) when the last block
        had to be copied first."""
        table = self.block_tables[seq.seq_id]
        logical = seq.num_logical_blocks
        if len(table) < logical:
            assert len(table) == logical - 1
            window = self.block_sliding_window
            if window and len(table) >= window:
                table.append(table[len(table) % window])
            else:
                table.append(self.gpu_allocator.allocate())
                return None
        last = table[-1]
        if self.gpu_allocator.ref_count(last) == 1:
            return None
        fresh = self.gpu_allocator.allocate()
        table[-1] = fresh
        self.gpu_allocator.free(last)
        return last, fresh

    def fork(self, parent_seq: Sequence, child_seq: Sequence) -> None:
        src = self.block_tables[parent_seq.seq_id]
        self.block_tables[child_seq.seq_id] = list(src)
        for block in dict.fromkeys(src):     # a sliding window repeats blocks: count each once
            self.gpu_allocator.share(block)

    # --------------------------------------------------------------------- swap
    def _distinct_blocks(self, seq_group: SequenceGroup) -> List[int]:
        seen: Dict[int, None] = {}
        for seq in seq_group.get_seqs():
            if not seq.is_finished():
                seen.update(dict.fromkeys(self.block_tables[seq.seq_id]))
        return list(seen)

    def can_swap_in(self, seq_group: SequenceGroup) -> bool:
        need = len(self._distinct_blocks(seq_group)) + seq_group.num_seqs(status=SequenceStatus.SWAPPED)
        return self.gpu_allocator.get_num_free_blocks() - need >= self.watermark_blocks

    def _move(self, seq_group: SequenceGroup, status: SequenceStatus, src: BlockPool,
              dst: BlockPool) -> Dict[int, int]:
        mapping: Dict[int, int] = {}
        for seq in seq_group.get_seqs(status=status):
            new_table = []
            for block in self.block_tables[seq.seq_id]:
                if block in mapping:
                    dst.share(mapping[block])
                else:
                    mapping[block] = dst.allocate()
                new_table.append(mapping[block])
                src.free(block)
            self.block_tables[seq.seq_id] = new_table
            if dst.device == Device.CPU:
                self._on_cpu.add(seq.seq_id)
            else:
                self._on_cpu.discard(seq.seq_id)
        return mapping

    def swap_in(self, seq_group: SequenceGroup) -> Dict[int, int]:
        """CPU block -> GPU block for every swapped sequence of the group."""
        return self._move(seq_group, SequenceStatus.SWAPPED, self.cpu_allocator, self.gpu_allocator)

    def can_swap_out(self, seq_group: SequenceGroup) -> bool:
        return len(self._distinct_blocks(seq_group)) <= self.cpu_allocator.get_num_free_blocks()

    def swap_out(self, seq_group: SequenceGroup) -> Dict[int, int]:
        """GPU block -> CPU block for every running sequence of the group."""
        return self._move(seq_group, SequenceStatus.RUNNING, self.gpu_allocator, self.cpu_allocator)

    # --------------------------------------------------------------------- free
    def _free_table(self, seq_id: int, table: List[int]) -> None:
        window = self.block_sliding_window
        blocks = table[-window:] if window is not None else table
        pool = self.cpu_allocator if seq_id in self._on_cpu else self.gpu_allocator
        for block in dict.fromkeys(blocks):
            pool.free(block)

    def free(self, seq: Sequence) -> None:
        table = self.block_tables.pop(seq.seq_id, None)
        if table is None:
            return                            # already freed, or never scheduled
        self._free_table(seq.seq_id, table)
        self._on_cpu.discard(seq.seq_id)

    def reset(self) -> None:
        for seq_id, table in self.block_tables.items():
            self._free_table(seq_id, table)
        self.block_tables.clear()
        self._on_cpu.clear()

    # ------------------------------------------------------------------ queries
    def get_block_table(self, seq: Sequence) -> List[int]:
        return list(self.block_tables[seq.seq_id])

    def get_num_free_gpu_blocks(self) -> int:
        return self.gpu_allocator.get_num_free_blocks()

    def get_num_free_cpu_blocks(self) -> int:
        return self.cpu_allocator.get_num_free_blocks()

    # prefix-caching hooks of the reference interface: no-ops with caching off
    def access_all_blocks_in_seq(self, seq: Sequence, access_time: float) -> None:
        pass

    def get_common_computed_block_ids(self, seq_group: SequenceGroup) -> List[int]:
        return []

    def mark_blocks_as_computed(self, seq_group: SequenceGroup) -> None:
        pass
