"""Paged KV-cache block accounting.

Behaviour restated from vllm/core/block_manager.py:13-487 with prefix caching off
(the allocator is then its FIFO variant, :35-38; evictor vllm/core/evictor.py:122-152):

  * physical block numbers are handed out 0, 1, 2, ... until every block of the pool
    has been used once; only then are freed blocks reused, oldest-freed first;
  * a prompt needs one block per started block_size tokens; an append needs a new
    block when the sequence grows into a new logical block;
  * a block shared by forked sequences is copied on the first append (copy-on-write);
  * watermark: a prompt is admitted only if 1% of the pool stays free afterwards.

With prefix caching on (``enable_caching``) the pools are ``CachingBlockPool``s: full blocks are
keyed by a hash of every token up to their end (vllm/sequence.py:188-197), a prompt whose leading
blocks are already in the pool shares them, freed blocks stay findable until they are evicted, and
eviction is least-recently-used with ties going to the block that hashes the most tokens
(vllm/core/evictor.py:68-104); ``computed`` marks blocks whose KV has been written, so that a later
prompt can skip them (block_manager.py:441-487).

Blocks are plain integers with per-pool side tables; there are no per-block objects.  When a sequence is freed its blocks return to the pool in block-table order
(the reference iterates a set of objects there, so its order depends on memory
addresses and cannot be reproduced; block numbering is identical to the reference's
until the first reuse of a freed block).
"""
import enum
from collections import deque
from itertools import count, takewhile
from os.path import commonprefix
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


class CachingBlockPool:
    """Free-block accounting of one device with content-addressed blocks: BlockAllocator with
    ``enable_caching=True`` (vllm/core/block_manager.py:13-113) over the LRU evictor
    (vllm/core/evictor.py:53-119).  Same method names as ``BlockPool``; ``allocate`` takes the hash."""

    def __init__(self, device: Device, block_size: int, num_blocks: int) -> None:
        self.device = device
        self.block_size = block_size
        self.num_blocks = num_blocks
        self.current_num_blocks = 0                    # blocks handed out at least once
        self.cached_blocks: Dict[int, int] = {}        # hash -> block, blocks in use
        self.free_table: Dict[int, int] = {}           # hash -> block, freed but not evicted (insertion order)
        self._ref = [0] * num_blocks
        self.block_hash: List[Optional[int]] = [None] * num_blocks
        self.num_hashed_tokens = [0] * num_blocks
        self.last_accessed = [-1.0] * num_blocks       # DEFAULT_LAST_ACCESSED_TIME, vllm/block.py:8
        self.computed = [False] * num_blocks
        self.default_hash_ctr = count()

    def _evict(self) -> int:
        """evictor.py:68-104: lowest last_accessed; among those the most hashed tokens; among those
        the one freed first."""
        if not self.free_table:
            raise ValueError("No usable cache memory left")
        best = None
        for block in self.free_table.values():
            if (best is None or self.last_accessed[block] < self.last_accessed[best]
                    or (self.last_accessed[block] == self.last_accessed[best]
                        and self.num_hashed_tokens[block] > self.num_hashed_tokens[best])):
                best = block
        del self.free_table[self.block_hash[best]]
        self.computed[best] = False
        return best

    def _allocate_block(self, block_hash: int, num_hashed_tokens: int) -> int:
        if self.current_num_blocks == self.num_blocks:
            block = self._evict()
        else:
            block = self.current_num_blocks
            self.current_num_blocks += 1
        self.block_hash[block] = block_hash
        self.num_hashed_tokens[block] = num_hashed_tokens
        return block

    def allocate(self, block_hash: Optional[int] = None, num_hashed_tokens: int = 0) -> int:
        if block_hash is None:
            block_hash = next(self.default_hash_ctr)
        if block_hash in self.free_table:              # freed, not evicted yet: bring it back
            assert block_hash not in self.cached_blocks
            block = self.free_table.pop(block_hash)
            assert self._ref[block] == 0
            self.cached_blocks[block_hash] = block
            self._ref[block] += 1
            return block
        if block_hash not in self.cached_blocks:
            self.cached_blocks[block_hash] = self._allocate_block(block_hash, num_hashed_tokens)
        block = self.cached_blocks[block_hash]
        self._ref[block] += 1
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
            h = self.block_hash[block]
            assert h not in self.free_table
            self.free_table[h] = block
            del self.cached_blocks[h]

    def get_num_free_blocks(self) -> int:
        return self.num_blocks - self.current_num_blocks + len(self.free_table)

    def contains_block(self, block_hash: int) -> bool:
        return block_hash in self.cached_blocks or block_hash in self.free_table

    def update_hash(self, block_hash: int, block: int) -> None:
        assert not self.contains_block(block_hash)
        del self.cached_blocks[self.block_hash[block]]
        self.block_hash[block] = block_hash
        self.cached_blocks[block_hash] = block


class AllocStatus(enum.Enum):
    """vllm/core/block_manager.py:116-127."""
    OK = enum.auto()
    LATER = enum.auto()
    NEVER = enum.auto()


class BlockSpaceManager:
    def __init__(self, block_size: int, num_gpu_blocks: int, num_cpu_blocks: int, watermark: float = 0.01,
                 sliding_window: Optional[int] = None, enable_caching: bool = False) -> None:
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
        self.enable_caching = enable_caching
        pool = CachingBlockPool if enable_caching else BlockPool
        self.gpu_allocator = pool(Device.GPU, block_size, num_gpu_blocks)
        self.cpu_allocator = pool(Device.CPU, block_size, num_cpu_blocks)
        self.block_tables: Dict[int, List[int]] = {}
        self._on_cpu: set = set()            # seq ids whose table holds CPU block numbers
        # Blocks that fill up during DECODE steps become shareable by content (see _maybe_promote_last_block).
        # With split pools their KV exists on the token pool only, so a later prompt that shared them would
        # have its prompt worker read blocks that were never written there: the scheduler turns this off.
        self.promote_decode_blocks = True
        # Pipelined steps make room for a token that has not been sampled yet (the sequence ends in a placeholder).
        # A block that becomes full that way has no content to be addressed by until the token arrives:
        # seq id -> index of that block in the sequence's table, hashed by the next append_slot (_promote_postponed).
        self._postponed: Dict[int, int] = {}

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
            elif self.enable_caching:
                # block_manager.py:199-201: every prompt block is addressed by the hash of the tokens up to
                # its end -- the last, partly filled one too (over tokens it does not hold yet)
                table.append(self.gpu_allocator.allocate(waiting[0].hash_of_block(logical),
                                                         waiting[0].num_hashed_tokens_of_block(logical)))
            else:
                table.append(self.gpu_allocator.allocate())
        # every waiting sequence of the group starts from the same prompt blocks; the
        # reference hands each a copy of the table without raising reference counts
        for seq in waiting:
            self.block_tables[seq.seq_id] = list(table)

    # ------------------------------------------------------------------ appends
    def can_append_slot(self, seq_group: SequenceGroup) -> bool:
        return seq_group.num_seqs(status=SequenceStatus.RUNNING) <= self.gpu_allocator.get_num_free_blocks()

    def _is_last_block_full(self, seq: Sequence) -> bool:
        n = seq.get_len()
        return n > 0 and n % self.block_size == 0

    def _allocate_last_block(self, seq: Sequence) -> int:
        """block_manager.py:254-267."""
        if not self.enable_caching:
            return self.gpu_allocator.allocate()
        last = seq.num_logical_blocks - 1
        block_hash = None
        if self._is_last_block_full(seq) and self.promote_decode_blocks:
            if seq.has_placeholder():
                self._postponed[seq.seq_id] = last       # its newest token is not known yet: no content hash
            else:
                block_hash = seq.hash_of_block(last)
        return self.gpu_allocator.allocate(block_hash, seq.num_hashed_tokens_of_block(last))

    def _maybe_promote_last_block(self, seq: Sequence, block: int) -> int:
        """A block that has just become full gets its content hash, so that other sequences can share it;
        if a block with that content exists already, that one is used instead (block_manager.py:218-252)."""
        if not self.enable_caching or not self.promote_decode_blocks or not self._is_last_block_full(seq):
            return block
        if seq.has_placeholder():
            # Hashing the placeholder would give every sequence with the same tokens BEFORE it the same block,
            # whatever each of them samples next.
            self._postponed[seq.seq_id] = seq.num_logical_blocks - 1
            return block
        new_hash = seq.hash_of_block(seq.num_logical_blocks - 1)
        if self.gpu_allocator.contains_block(new_hash):
            self.gpu_allocator.free(block)
            return self.gpu_allocator.allocate(new_hash)
        self.gpu_allocator.update_hash(new_hash, block)
        return block

    def _promote_postponed(self, seq: Sequence) -> None:
        """Gives a block that filled up under a placeholder its content hash once the token is there.  The
        block keeps its place (other steps in flight already address it): if a block with that content exists
        the two are not merged, this one just stays private."""
        index = self._postponed.get(seq.seq_id)
        if index is None:
            return
        known = seq.get_len() - (1 if seq.has_placeholder() else 0)
        if known < (index + 1) * self.block_size:
            return
        del self._postponed[seq.seq_id]
        table = self.block_tables[seq.seq_id]
        if seq.seq_id in self._on_cpu or self.block_sliding_window is not None or index >= len(table):
            return
        block, new_hash = table[index], seq.hash_of_block(index)
        if self.gpu_allocator.ref_count(block) == 1 and not self.gpu_allocator.contains_block(new_hash):
            self.gpu_allocator.update_hash(new_hash, block)

    def append_slot(self, seq: Sequence) -> Optional[Tuple[int, int]]:
        """Makes room for the sequence's newest token; returns (src, dst) when the last block
        had to be copied first."""
        table = self.block_tables[seq.seq_id]
        if self._postponed:
            self._promote_postponed(seq)
        logical = seq.num_logical_blocks
        if len(table) < logical:
            assert len(table) == logical - 1
            window = self.block_sliding_window
            if window and len(table) >= window:
                table.append(table[len(table) % window])
            else:
                table.append(self._allocate_last_block(seq))
                return None
        last = table[-1]
        if self.gpu_allocator.ref_count(last) == 1:
            table[-1] = self._maybe_promote_last_block(seq, last)
            return None
        fresh = self._allocate_last_block(seq)
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
                elif self.enable_caching:
                    mapping[block] = dst.allocate(src.block_hash[block], src.num_hashed_tokens[block])
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
        self._postponed.pop(seq.seq_id, None)

    def reset(self) -> None:
        for seq_id, table in self.block_tables.items():
            self._free_table(seq_id, table)
        self.block_tables.clear()
        self._on_cpu.clear()
        self._postponed.clear()

    # ------------------------------------------------------------------ queries
    def get_block_table(self, seq: Sequence) -> List[int]:
        return list(self.block_tables[seq.seq_id])

    def get_num_free_gpu_blocks(self) -> int:
        return self.gpu_allocator.get_num_free_blocks()

    def get_num_free_cpu_blocks(self) -> int:
        return self.cpu_allocator.get_num_free_blocks()

    # ------------------------------------------------------------ prefix caching
    def access_all_blocks_in_seq(self, seq: Sequence, access_time: float) -> None:
        """block_manager.py:432-439 (the evictor's clock)."""
        if not self.enable_caching:
            return
        stamps = self.gpu_allocator.last_accessed
        for block in self.block_tables[seq.seq_id]:
            stamps[block] = access_time

    def compute_full_blocks_in_seq(self, seq: Sequence) -> None:
        """block_manager.py:441-451, as written there: of the blocks that the sequence fills completely, all
        but the last are marked, walking down until a marked one is met."""
        table = self.block_tables.get(seq.seq_id)
        if table is None:
            return
        max_full_block = seq.get_len() // self.block_size - 1
        if max_full_block == -1:
            return
        computed = self.gpu_allocator.computed
        for i in reversed(range(max_full_block)):
            if computed[table[i]]:
                break
            computed[table[i]] = True

    def get_all_computed_blocks(self, seq: Sequence) -> List[int]:
        """Leading computed blocks, never the last block of the table (a prompt that is cached completely
        would leave the model nothing to run, block_manager.py:453-464)."""
        table = self.block_tables.get(seq.seq_id)
        if table is None:
            return []
        computed = self.gpu_allocator.computed
        return list(takewhile(lambda b: computed[b], table[:-1]))

    def get_common_computed_block_ids(self, seq_group: SequenceGroup) -> List[int]:
        if not self.enable_caching:
            return []
        ids_list = [self.get_all_computed_blocks(seq) for seq in seq_group.get_seqs()]
        # (os.path.commonprefix of no lists is the empty STRING, which the reference passes on; a list here)
        return list(commonprefix([ids for ids in ids_list if ids != []]))

    def mark_blocks_as_computed(self, seq_group: SequenceGroup) -> None:
        if self.enable_caching:
            for seq in seq_group.get_seqs():
                self.compute_full_blocks_in_seq(seq)
