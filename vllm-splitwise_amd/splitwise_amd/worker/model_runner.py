"""Turns scheduled sequence groups into device tensors and runs the model.

Index arithmetic restated from vllm/worker/model_runner.py:126-399: slot of token i of a
sequence = block_table[i // block_size] * block_size + i % block_size; padding slots are -1;
a decode step feeds the last token at position len - 1 with context_len = len (capped by the
sliding window).  Two prompt layouts are produced:

  * ``_prepare_prompt``   the reference's padded [num_prompts, max_prompt_len] batch
                          (kept bit-identical; used by the parity tests and by callers that
                          expect the reference's shapes);
  * ``_prepare_prompt_packed``  the layout the engine runs: prompts back to back in one
                          token dimension, each starting at a multiple of 16 rows so the KV
                          write lands in whole cache-block pieces; no compute is spent on
                          per-prompt padding to the longest prompt.

Decode steps are replayed from HIP graphs captured per padded batch size
(vllm/worker/model_runner.py:701-876 does the same with CUDA graphs); the hand-off wait of
the token pool stays outside the graph (vllm/worker/worker.py:318-325).
"""
import time
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from ..config import DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig
from ..model_executor.input_metadata import InputMetadata
from ..model_executor.layers.sampler import FollowerSamples, PendingSamples, SamplingMetadata
from ..model_executor.model_loader import get_model
from ..sampling_params import SamplingParams
from ..sequence import PLACEHOLDER_TOKEN_ID, SamplerOutput, SequenceData, SequenceGroupMetadata

_PAD_SLOT_ID = -1
_PROMPT_ROW_ALIGN = 16
# decode batch sizes with a captured graph: 1, 2, 4, 8, 16, 24, ..., 256
_BATCH_SIZES_TO_CAPTURE = [1, 2, 4] + [8 * i for i in range(1, 33)]


def _get_graph_batch_size(batch_size: int) -> int:
    if batch_size <= 2:
        return batch_size
    if batch_size <= 4:
        return 4
    return (batch_size + 7) // 8 * 8


def _pad_2d(rows: List[List[int]], width: int, pad: int, dtype) -> np.ndarray:
    out = np.full((len(rows), width), pad, dtype=dtype)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out


class ModelRunner:
    def __init__(self, model_config: Optional[ModelConfig], parallel_config: Optional[ParallelConfig],
                 scheduler_config: Optional[SchedulerConfig], device_config: Optional[DeviceConfig],
                 kv_cache_dtype: str = "auto", is_driver_worker: bool = False) -> None:
        self.model_config = model_config
        self.parallel_config = parallel_config
        self.scheduler_config = scheduler_config
        self.device_config = device_config if device_config is not None else DeviceConfig("cpu")
        self.device = self.device_config.device
        self.is_driver_worker = is_driver_worker
        self.driver_rank = 0
        self.sliding_window = model_config.get_sliding_window() if model_config is not None else None
        self.kv_cache_dtype = kv_cache_dtype
        self.model = None
        self.block_size: Optional[int] = None
        self.graph_runners: Dict[int, "HIPGraphRunner"] = {}
        # batch sizes that were also captured with the partitioned decode-attention kernel (see capture_model)
        self.graph_runners_partitioned: Dict[int, "HIPGraphRunner"] = {}
        self.graph_memory_pool = None
        self.max_context_len_to_capture = model_config.max_context_len_to_capture if model_config else 0
        self.enforce_eager = model_config.enforce_eager if model_config is not None else True
        # samples of the previous step that were left on the device (pipelined engine steps)
        self.pending: Optional[PendingSamples] = None
        self._generators: Dict[str, torch.Generator] = {}

    # ------------------------------------------------------------------ set-up
    def load_model(self) -> None:
        self.model = get_model(self.model_config, self.device)

    def set_block_size(self, block_size: int) -> None:
        self.block_size = block_size

    @property
    def vocab_size(self) -> int:
        return self.model_config.get_vocab_size()

    def _to_device(self, array: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(array)).to(dtype)
        if self.device.type == "cuda":
            return t.pin_memory().to(self.device, non_blocking=True)
        return t

    # ----------------------------------------------------------------- prompts
    def _prompt_rows(self, metas: List[SequenceGroupMetadata]):
        """Per prompt: (new tokens, their positions, their slots, tokens already cached, blocks of those).
        Leading blocks that the block manager reports as computed are skipped: only the rest of the prompt
        is run, against the cached context (model_runner.py:153-168; prefix caching)."""
        rows = []
        for meta in metas:
            assert meta.is_prompt
            seq_ids = list(meta.seq_data.keys())
            assert len(seq_ids) == 1
            seq_id = seq_ids[0]
            tokens = meta.seq_data[seq_id].get_token_ids()
            n = len(tokens)
            computed, prefix = 0, []
            if meta.computed_block_nums and self.sliding_window is None and meta.block_tables is not None:
                prefix = list(meta.computed_block_nums)
                computed = len(prefix) * self.block_size
            if meta.block_tables is None:              # profiling run: nothing is cached
                slots = [_PAD_SLOT_ID] * n
            else:
                table = meta.block_tables[seq_id]
                bs = self.block_size
                # with a sliding window only the last `window` tokens are kept (model_runner.py:195-209)
                start = 0 if self.sliding_window is None else max(0, n - self.sliding_window)
                slots = [_PAD_SLOT_ID if i < start else table[i // bs] * bs + i % bs for i in range(computed, n)]
            rows.append((tokens[computed:], list(range(computed, n)), slots, computed, prefix))
        return rows

    def _prefix_tensors(self, rows):
        """(cached lengths, block tables of the cached prefixes) as the prompt kernels take them; the table
        has no columns when no prompt of the step has a cached prefix (model_runner.py:236-249)."""
        context_lens = self._to_device(np.asarray([r[3] for r in rows], dtype=np.int32), torch.int32)
        width = max(len(r[4]) for r in rows)
        if width == 0:
            return context_lens, torch.empty((len(rows), 0), dtype=torch.int32, device=self.device)
        return context_lens, self._to_device(_pad_2d([r[4] for r in rows], width, 0, np.int32), torch.int32)

    def _prepare_prompt(self, metas: List[SequenceGroupMetadata]):
        """Padded layout, identical to the reference's tensors (model_runner.py:126-272).
        -> (input_tokens [B, L], input_positions [B, L], InputMetadata, prompt_lens, subquery_lens)"""
        assert len(metas) > 0
        rows = self._prompt_rows(metas)
        subquery_lens = [len(r[0]) for r in rows]
        prompt_lens = [r[3] + len(r[0]) for r in rows]
        max_len = max(subquery_lens)
        assert max_len > 0
        tokens = self._to_device(_pad_2d([r[0] for r in rows], max_len, 0, np.int64), torch.long)
        positions = self._to_device(_pad_2d([r[1] for r in rows], max_len, 0, np.int64), torch.long)
        slots = self._to_device(_pad_2d([r[2] for r in rows], max_len, _PAD_SLOT_ID, np.int64), torch.long)
        start_loc = self._to_device(np.arange(0, len(rows) * max_len, max_len, dtype=np.int32), torch.int32)
        context_lens, block_tables = self._prefix_tensors(rows)
        md = InputMetadata(
            is_prompt=True, slot_mapping=slots,
            prompt_lens=self._to_device(np.asarray(prompt_lens, dtype=np.int32), torch.int32),
            max_seq_len=max_len, start_loc=start_loc, max_context_len=None,
            context_lens=context_lens, block_tables=block_tables,
            use_cuda_graph=False, kv_cache_dtype=self.kv_cache_dtype)
        return tokens, positions, md, prompt_lens, subquery_lens

    def _prepare_prompt_packed(self, metas: List[SequenceGroupMetadata]):
        """-> (input_tokens [T], input_positions [T], InputMetadata, subquery_lens, row starts)."""
        rows = self._prompt_rows(metas)
        subquery_lens = [len(r[0]) for r in rows]
        prompt_lens = [r[3] + len(r[0]) for r in rows]
        starts, total = [], 0
        for n in subquery_lens:
            starts.append(total)
            total += -(-n // _PROMPT_ROW_ALIGN) * _PROMPT_ROW_ALIGN
        tokens = np.zeros(total, dtype=np.int64)
        positions = np.zeros(total, dtype=np.int64)
        slots = np.full(total, _PAD_SLOT_ID, dtype=np.int64)
        for (tok, pos, slot, _, _), s in zip(rows, starts):
            n = len(tok)
            tokens[s:s + n] = tok
            positions[s:s + n] = pos
            slots[s:s + n] = slot
        context_lens = block_tables = None
        if any(r[3] for r in rows):
            context_lens, block_tables = self._prefix_tensors(rows)
        md = InputMetadata(
            is_prompt=True, slot_mapping=self._to_device(slots, torch.long),
            prompt_lens=self._to_device(np.asarray(prompt_lens, dtype=np.int32), torch.int32),
            max_seq_len=max(subquery_lens), start_loc=self._to_device(np.asarray(starts, dtype=np.int32), torch.int32),
            max_context_len=None, context_lens=context_lens, block_tables=block_tables, use_cuda_graph=False,
            kv_cache_dtype=self.kv_cache_dtype)
        return (self._to_device(tokens, torch.long), self._to_device(positions, torch.long), md, subquery_lens,
                starts)

    # ------------------------------------------------------------------ decode
    def _prepare_decode(self, metas: List[SequenceGroupMetadata]):
        """-> (input_tokens [B, 1], input_positions [B, 1], InputMetadata); model_runner.py:274-399."""
        assert len(metas) > 0
        tokens: List[int] = []
        positions: List[int] = []
        slots: List[int] = []
        context_lens: List[int] = []
        tables: List[List[int]] = []
        # row of the previous step's device-resident samples that holds a sequence's last token when
        # the host does not know it yet; -1 where the host's value is the one to use
        from_pending: List[int] = []
        bs = self.block_size
        for meta in metas:
            assert not meta.is_prompt
            for seq_id, data in meta.seq_data.items():
                seq_len = data.get_len()
                pos = seq_len - 1
                last = data.get_last_token_id()
                if last == PLACEHOLDER_TOKEN_ID:
                    if self.pending is None or seq_id not in self.pending.row_of:
                        raise RuntimeError(f"sequence {seq_id} waits for a token no earlier step left on the device")
                    from_pending.append(self.pending.row_of[seq_id])
                    last = 0
                else:
                    from_pending.append(-1)
                tokens.append(last)
                positions.append(pos)
                context_lens.append(seq_len if self.sliding_window is None else min(seq_len, self.sliding_window))
                table = meta.block_tables[seq_id]
                slots.append(table[pos // bs] * bs + pos % bs)
                if self.sliding_window is not None:
                    table = table[-(self.sliding_window // bs):]
                tables.append(table)
        batch = len(tokens)
        max_ctx = max(context_lens)
        use_graph = (not self.enforce_eager and bool(self.graph_runners)
                     and batch <= _BATCH_SIZES_TO_CAPTURE[-1] and max_ctx <= self.max_context_len_to_capture)
        if use_graph:
            padded = _get_graph_batch_size(batch)
            for _ in range(padded - batch):
                tokens.append(0)
                from_pending.append(-1)
                positions.append(0)
                slots.append(_PAD_SLOT_ID)
                context_lens.append(1)
                tables.append([])
            width = -(-self.max_context_len_to_capture // bs)
        else:
            width = max(len(t) for t in tables)
        md = InputMetadata(
            is_prompt=False, slot_mapping=self._to_device(np.asarray(slots, dtype=np.int64).reshape(-1, 1), torch.long),
            prompt_lens=None, max_seq_len=None, start_loc=None, max_context_len=max_ctx,
            context_lens=self._to_device(np.asarray(context_lens, dtype=np.int32), torch.int32),
            block_tables=self._to_device(_pad_2d(tables, width, 0, np.int32), torch.int32),
            use_cuda_graph=use_graph, kv_cache_dtype=self.kv_cache_dtype,
            mean_context_len=None if use_graph else sum(context_lens) / batch)
        if use_graph and padded in self.graph_runners_partitioned:
            # this batch size was captured with either decode-attention kernel (the rule picked the unpartitioned one
            # for the capture's maximum context): a ragged batch replays the partitioned form
            real = context_lens[:batch]
            if not self._decode_rule(batch, max(real), sum(real) / batch):
                md.decode_kernel = "v2"
        token_tensor = self._to_device(np.asarray(tokens, dtype=np.int64).reshape(-1, 1), torch.long)
        if any(r >= 0 for r in from_pending):
            src = self._to_device(np.asarray(from_pending, dtype=np.int64), torch.long)
            on_device = self.pending.next_ids.index_select(0, src.clamp(min=0)).view(-1, 1)
            token_tensor = torch.where((src >= 0).view(-1, 1), on_device, token_tensor)
        return (token_tensor, self._to_device(np.asarray(positions, dtype=np.int64).reshape(-1, 1), torch.long), md)

    # ---------------------------------------------------------------- sampling
    def _prepare_sample(self, metas: List[SequenceGroupMetadata], last_token_rows: Optional[List[int]]):
        """Rows of the hidden states to sample from: the last token of every prompt, or every row
        of a decode step (model_runner.py:401-485 without prompt_logprobs)."""
        seq_groups: List[Tuple[List[int], SamplingParams]] = []
        selected: List[int] = []
        generators: Dict[int, torch.Generator] = {}
        seq_data: Dict[int, SequenceData] = {}
        row = 0
        for i, meta in enumerate(metas):
            seq_ids = list(meta.seq_data.keys())
            params = meta.sampling_params
            seq_groups.append((seq_ids, params))
            seq_data.update(meta.seq_data)
            if meta.is_prompt:
                selected.append(last_token_rows[i])
                n_rows = 1
            else:
                selected.extend(range(row, row + len(seq_ids)))
                n_rows = len(seq_ids)
            if params.seed is not None:
                if meta.state.generator is None:
                    # the state object of a command that crossed a process boundary is a fresh copy:
                    # the request's generator is kept here, by request id
                    gen = self._generators.get(meta.request_id)
                    if gen is None or meta.is_prompt:
                        gen = torch.Generator(device=self.device).manual_seed(params.seed)
                        if len(self._generators) >= 4096:
                            self._generators.pop(next(iter(self._generators)))
                        self._generators[meta.request_id] = gen
                    meta.state.generator = gen
                for r in range(len(selected) - n_rows, len(selected)):
                    generators[r] = meta.state.generator
            row += n_rows
        return SamplingMetadata(seq_groups, seq_data, None,
                                self._to_device(np.asarray(selected, dtype=np.int64), torch.long), generators,
                                is_prompt=[meta.is_prompt for meta in metas])

    # --------------------------------------------------------------- execution
    @torch.inference_mode()
    def execute_model(self, metas: List[SequenceGroupMetadata], kv_caches, blocks_to_nw=None,
                      prompt_layout: str = "packed", defer: bool = False, nw_peer: int = 0,
                      share_samples: Optional[bool] = None, track_pending: bool = True):
        """``defer``: return the step's ``PendingSamples`` (tokens still on the device) instead of a
        ``SamplerOutput``; the next decode step takes unknown last tokens from it.  ``share_samples``
        (tensor-parallel pools; default: whenever ``defer``): send the sampled ids to the pool's other ranks,
        which must all make this call the same way -- off for a prompt pool, whose ranks never decode.
        ``track_pending`` off: the step's samples are not what the next decode step continues from (a prompt step
        that runs beside the decode steps, on a stream of its own)."""
        is_prompt = metas[0].is_prompt
        restore = None
        if is_prompt:
            if prompt_layout == "packed":
                tokens, positions, md, prompt_lens, starts = self._prepare_prompt_packed(metas)
                last_rows = [s + n - 1 for s, n in zip(starts, prompt_lens)]
            else:
                tokens, positions, md, _, prompt_lens = self._prepare_prompt(metas)     # rows = the new tokens
                width = tokens.shape[1]
                last_rows = [i * width + n - 1 for i, n in enumerate(prompt_lens)]
            sampling_md = self._prepare_sample(metas, last_rows)
        else:
            restore = self._longest_first(metas)
            if restore is not None:
                metas = [metas[i] for i in restore[0]]
            tokens, positions, md = self._prepare_decode(metas)
            sampling_md = self._prepare_sample(metas, None)
        md.blocks_to_nw = blocks_to_nw or {}
        md.nw_peer = nw_peer
        # the model sees one flat token dimension in every layout
        if md.use_cuda_graph:
            runners = self.graph_runners_partitioned if md.decode_kernel == "v2" else self.graph_runners
            hidden = runners[tokens.shape[0]](tokens, positions, kv_caches, md)
        else:
            hidden = self.model(tokens.view(-1), positions.view(-1), kv_caches, md)
        sampling_md.defer = defer
        result = self.model.sample(hidden, sampling_md)
        if share_samples is None:
            share_samples = defer
        if share_samples and self.parallel_config is not None and self.parallel_config.tensor_parallel_size > 1:
            result = self._share_sampled_ids(result, sampling_md)
        if not is_prompt and restore is not None and result is not None:
            result.reorder(restore[1])           # report in the order the step was asked for
        if track_pending:
            self.pending = result if defer else None
        return result

    def _share_sampled_ids(self, result, sampling_md: SamplingMetadata):
        """Pipelined steps of a tensor-parallel pool: only the pool's first rank samples (the logits are gathered
        there), but every rank's NEXT decode step reads its input tokens from the device.  The first rank sends
        the sampled ids to the others, device to device; they keep them as ``FollowerSamples``."""
        from ..model_executor.parallel_state import tensor_model_parallel_broadcast_from_first
        rows = sampling_md.selected_token_indices.numel()
        if result is not None:
            ids = result.next_ids
            if ids.dtype != torch.long or not ids.is_contiguous():
                ids = ids.to(torch.long).contiguous()
            tensor_model_parallel_broadcast_from_first(ids)
            return result
        ids = torch.empty(rows, dtype=torch.long, device=self.device)
        tensor_model_parallel_broadcast_from_first(ids)
        return FollowerSamples(sampling_md.seq_groups, ids)

    def _decode_rule(self, num_seqs: int, max_ctx: int, mean_ctx: Optional[float]) -> bool:
        """PagedAttentionImpl.use_v1 for this model's attention geometry."""
        from ..model_executor.layers.attention import PagedAttentionImpl
        tp = self.parallel_config.tensor_parallel_size
        heads = self.model_config.hf_config["num_attention_heads"] // tp
        kv_heads = self.model_config.get_num_kv_heads(self.parallel_config)
        width = PagedAttentionImpl.query_group_width(heads, kv_heads, self.model_config.get_head_size(), self.block_size,
                                                     self.model_config.dtype, self.kv_cache_dtype)
        return PagedAttentionImpl.use_v1(num_seqs, heads, kv_heads, max_ctx, width, mean_ctx)

    # a decode batch is reordered when it has at least this many sequences and its longest context is this many
    # times the mean
    sort_decode_min_batch = 16
    sort_decode_min_ratio = 1.5

    def _longest_first(self, metas: List[SequenceGroupMetadata]):
        """-> (order, inverse) to run a ragged decode batch longest context first, or None to leave it alone.
        The decode-attention kernels start their workgroups in batch order, and an unpartitioned workgroup walks
        one sequence's whole context: a long context at the END of the batch starts last and finishes long after
        everything else (6 sequences of 6000 tokens behind 250 of 600: 582 us against 516 us the other way
        round).  Nothing else in a step depends on the order of its rows; the samples are put back into the
        order of ``metas`` before they are reported."""
        n = len(metas)
        if n < self.sort_decode_min_batch or any(len(m.seq_data) != 1 for m in metas):
            return None
        lens = [next(iter(m.seq_data.values())).get_len() for m in metas]
        if max(lens) * n < self.sort_decode_min_ratio * sum(lens):
            return None
        order = sorted(range(n), key=lambda i: -lens[i])          # stable: equal lengths keep their order
        if order == list(range(n)):
            return None
        inverse = [0] * n
        for pos, i in enumerate(order):
            inverse[i] = pos
        return order, inverse

    @torch.inference_mode()
    def profile_run(self) -> None:
        """One prompt step of the largest admissible batch without a KV cache, to find the peak
        activation memory (model_runner.py:620-672)."""
        params = SamplingParams(temperature=0.0, max_tokens=1)
        max_tokens = self.scheduler_config.max_num_batched_tokens
        max_seqs = self.scheduler_config.max_num_seqs
        metas = []
        for gid in range(max_seqs):
            n = max_tokens // max_seqs + (gid < max_tokens % max_seqs)
            if n == 0:
                continue
            metas.append(SequenceGroupMetadata(str(gid), True, {gid: SequenceData([0] * n)}, params, None))
        num_layers = self.model_config.get_num_layers(self.parallel_config)
        self.execute_model(metas, [(None, None)] * num_layers)
        if self.device.type == "cuda":
            torch.cuda.synchronize()

    # --------------------------------------------------------------- HIP graphs
    @torch.inference_mode()
    def capture_model(self, kv_caches) -> None:
        """Captures one decode graph per batch size, largest first so that the smaller ones reuse
        its memory pool (model_runner.py:701-760)."""
        assert not self.enforce_eager
        t0 = time.perf_counter()
        max_bs = max(_BATCH_SIZES_TO_CAPTURE)
        max_blocks = -(-self.max_context_len_to_capture // self.block_size)
        dev = self.device
        tokens = torch.zeros(max_bs, 1, dtype=torch.long, device=dev)
        positions = torch.zeros(max_bs, 1, dtype=torch.long, device=dev)
        slots = torch.full((max_bs, 1), _PAD_SLOT_ID, dtype=torch.long, device=dev)
        context_lens = torch.ones(max_bs, dtype=torch.int32, device=dev)
        block_tables = torch.zeros(max_bs, max_blocks, dtype=torch.int32, device=dev)
        limit = min(max_bs, _get_graph_batch_size(self.scheduler_config.max_num_seqs))
        for bs in reversed([b for b in _BATCH_SIZES_TO_CAPTURE if b <= limit]):
            md = InputMetadata(is_prompt=False, slot_mapping=slots[:bs], prompt_lens=None, max_seq_len=None,
                               start_loc=None, max_context_len=self.max_context_len_to_capture,
                               context_lens=context_lens[:bs], block_tables=block_tables[:bs],
                               use_cuda_graph=True, kv_cache_dtype=self.kv_cache_dtype)
            runner = HIPGraphRunner(self.model)
            runner.capture(tokens[:bs], positions[:bs], kv_caches, md, self.graph_memory_pool)
            self.graph_memory_pool = runner.graph.pool()
            self.graph_runners[bs] = runner
            if (self._decode_rule(bs, self.max_context_len_to_capture, None)
                    and not self._decode_rule(bs, self.max_context_len_to_capture, 1.0)):
                # The rule takes the unpartitioned kernel for this launch but would partition a ragged batch of
                # the same size: a second graph with the partitioned kernel, same static buffers.
                md2 = InputMetadata(is_prompt=False, slot_mapping=slots[:bs], prompt_lens=None, max_seq_len=None,
                                    start_loc=None, max_context_len=self.max_context_len_to_capture,
                                    context_lens=context_lens[:bs], block_tables=block_tables[:bs],
                                    use_cuda_graph=True, kv_cache_dtype=self.kv_cache_dtype, decode_kernel="v2")
                second = HIPGraphRunner(self.model)
                second.capture(tokens[:bs], positions[:bs], kv_caches, md2, self.graph_memory_pool)
                self.graph_runners_partitioned[bs] = second
        self.capture_seconds = time.perf_counter() - t0


class HIPGraphRunner:
    """One captured decode step: static input buffers, graph replay, static output."""

    def __init__(self, model) -> None:
        self.model = model
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.inputs: Dict[str, torch.Tensor] = {}
        self.output: Optional[torch.Tensor] = None
        self.replays = 0

    def capture(self, tokens, positions, kv_caches, md: InputMetadata, pool) -> None:
        assert self.graph is None
        # warm-up run outside the capture (lazy initialisation inside libraries)
        self.model(tokens.view(-1), positions.view(-1), kv_caches, md)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=pool):
            hidden = self.model(tokens.view(-1), positions.view(-1), kv_caches, md)
        torch.cuda.synchronize()
        self.inputs = {"tokens": tokens, "positions": positions, "slot_mapping": md.slot_mapping,
                       "context_lens": md.context_lens, "block_tables": md.block_tables}
        self.output = hidden

    def __call__(self, tokens, positions, kv_caches, md: InputMetadata) -> torch.Tensor:
        del kv_caches                                   # baked into the graph
        self.inputs["tokens"].copy_(tokens, non_blocking=True)
        self.inputs["positions"].copy_(positions, non_blocking=True)
        self.inputs["slot_mapping"].copy_(md.slot_mapping, non_blocking=True)
        self.inputs["context_lens"].copy_(md.context_lens, non_blocking=True)
        self.inputs["block_tables"].copy_(md.block_tables, non_blocking=True)
        self.graph.replay()
        self.replays += 1
        return self.output
