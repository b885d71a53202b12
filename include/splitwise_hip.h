/*
 * splitwise_hip.h -- C ABI of libsplitwise_hip.so (gfx950 / MI355X).
 *
 * One `extern "C"` entry point per native operator that the reference binds
 * through pybind (`csrc/pybind.cpp:6-129`) or JIT-compiles for its KV transport
 * (`vllm/worker/comm_utils.py:22-87`).  Signatures carry raw device pointers,
 * explicit shapes / strides (in ELEMENTS unless a name ends in `_bytes`), dtype
 * enums and the HIP stream to enqueue on.  No torch types.
 *
 * Conventions
 *   - every function returns SW_OK (0) or a negative sw_status; the message of
 *     the last failure on the calling thread is returned by sw_last_error().
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*,
 *     NULL = the default stream) on the CURRENT device of the calling thread;
 *     nothing synchronises unless the comment says so.
 *   - tensors are borrowed for the duration of the call only; the caller
 *     allocates every output.
 *   - thread safe: no global mutable state except the per-thread error string.
 */
#ifndef SPLITWISE_HIP_H_
#define SPLITWISE_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sw_status {
  SW_OK = 0,
  SW_ERR_INVALID = -1,     /* unsupported head size / block size / dtype / bad shape:
                              the reference raises RuntimeError via TORCH_CHECK
                              (csrc/attention/attention_kernels.cu:675,707,732,745) */
  SW_ERR_HIP = -2,         /* a HIP runtime call failed */
  SW_ERR_TIMEOUT = -3      /* host-side wait ran out (only sw_kv_flag_wait_host) */
} sw_status;

typedef enum sw_dtype {
  SW_F32 = 0,
  SW_F16 = 1,
  SW_BF16 = 2,
  SW_FP8_E5M2 = 3          /* unscaled e5m2 stored as uint8 (kv_cache_dtype "fp8_e5m2") */
} sw_dtype;

const char* sw_last_error(void);
/* Version of this interface; sw_abi_version() returns the one the library was built from.
 * 2: sw_awq_to_kmajor, sw_int4_kmajor_gemm, sw_int4_kmajor_dequant added.
 * 3: sw_paged_attention_v2_static, sw_paged_attention_v2_static_max_bytes added.
 * 4: sw_kv_push takes max_workgroups.
 * 5: sw_rotary_embedding_and_cache added; the all-reduce signal area (sw_custom_ar_signal_bytes)
 *    grew from 2304 to 2560 bytes, which moves the status word at its end.
 * 6: sw_kv_push_mapped added.
 * 7: sw_context_attention_split, sw_context_attention_split_parts, sw_context_attention_split_workspace_bytes
 *    added. */
#define SW_ABI_VERSION 7
int sw_abi_version(void);

/* csrc/cuda_utils_kernels.cu:16-35 get_max_shared_memory_per_block_device_attribute:
 * LDS bytes one workgroup may use on `device`. */
int sw_get_max_shared_memory_per_block(int device, int* out_bytes);

/* ------------------------------------------------------------------ cache ops
 * csrc/cache_kernels.cu:228-268 reshape_and_cache.
 *   key/value      [num_tokens, num_heads, head_size] act_dtype, token stride
 *                  key_stride / value_stride (sliced out of a fused qkv tensor)
 *   key_cache      [num_blocks, num_heads, head_size/x, block_size, x]
 *   value_cache    [num_blocks, num_heads, head_size, block_size]
 *   slot_mapping   int64 [num_tokens]; negative slot = padding, skipped
 *   cache_dtype    == act_dtype ("auto") or SW_FP8_E5M2
 *   x              16 / sizeof(cache element)                              */
int sw_reshape_and_cache(const void* key, const void* value, void* key_cache,
                         void* value_cache, const int64_t* slot_mapping,
                         int num_tokens, int num_heads, int head_size,
                         int block_size, int x, int64_t key_stride,
                         int64_t value_stride, int act_dtype, int cache_dtype,
                         void* stream);

/* csrc/cache_kernels.cu:95-150 copy_blocks.  For every layer l and every pair p:
 * block dst[p] of key_cache_ptrs[l] and of value_cache_ptrs[l] := block src[p].
 *   key_cache_ptrs / value_cache_ptrs  DEVICE arrays of num_layers device pointers
 *   block_mapping                      DEVICE int64 [num_pairs][2] = (src, dst)
 *   block_bytes                        bytes of one block of one tensor, multiple of 16 */
int sw_copy_blocks(const void* const* key_cache_ptrs,
                   const void* const* value_cache_ptrs, int num_layers,
                   const int64_t* block_mapping, int num_pairs,
                   int64_t block_bytes, void* stream);

/* csrc/cache_kernels.cu:21-60 swap_blocks.  HOST arrays src_blocks/dst_blocks of
 * n entries; one async copy per maximal run of consecutive (src,dst) numbers.
 *   kind: 0 = device->device (same GPU), 1 = device->host, 2 = host->device  */
int sw_swap_blocks(const void* src, void* dst, const int64_t* src_blocks,
                   const int64_t* dst_blocks, int n, int64_t block_bytes,
                   int kind, void* stream);

/* csrc/cache_kernels.cu:296-320 convert_fp8_e5m2: elementwise cast of `numel`
 * elements.  Exactly one of src_dtype / dst_dtype is SW_FP8_E5M2.
 * to-fp8 is round-to-nearest-even, saturating to +-57344; NaN -> sign|0x7F
 * (csrc/quantization/fp8_e5m2_kvcache/quant_utils.cuh:164-192).             */
int sw_convert_fp8_e5m2(const void* src, void* dst, int64_t numel,
                        int src_dtype, int dst_dtype, void* stream);

/* ------------------------------------------------------------ paged attention
 * csrc/attention/attention_kernels.cu:711-747 paged_attention_v1.
 *   out            [num_seqs, num_heads, head_size] act_dtype, contiguous
 *   query          [num_seqs, num_heads, head_size], seq stride q_stride
 *   key_cache / value_cache as above; kv_block_stride / kv_head_stride in elements
 *   block_tables   int32 [num_seqs, max_num_blocks_per_seq]
 *   context_lens   int32 [num_seqs]
 *   alibi_slopes   float [num_heads] or NULL
 * head_size in {64,80,96,112,128,256}, block_size in {8,16,32}.            */
int sw_paged_attention_v1(void* out, const void* query, const void* key_cache,
                          const void* value_cache, int num_seqs, int num_heads,
                          int num_kv_heads, int head_size, float scale,
                          const int32_t* block_tables,
                          const int32_t* context_lens, int block_size,
                          int max_context_len, int max_num_blocks_per_seq,
                          const float* alibi_slopes, int64_t q_stride,
                          int64_t kv_block_stride, int64_t kv_head_stride,
                          int act_dtype, int cache_dtype, void* stream);

/* csrc/attention/attention_kernels.cu:901-940 paged_attention_v2: the context is
 * split into partitions of 512 tokens.
 *   exp_sums, max_logits  float [num_seqs, num_heads, max_num_partitions]
 *   tmp_out               act_dtype [num_seqs, num_heads, max_num_partitions, head_size]
 *   max_num_partitions    = ceil(max_context_len / 512)                   */
int sw_paged_attention_v2(void* out, float* exp_sums, float* max_logits,
                          void* tmp_out, const void* query,
                          const void* key_cache, const void* value_cache,
                          int num_seqs, int num_heads, int num_kv_heads,
                          int head_size, float scale,
                          const int32_t* block_tables,
                          const int32_t* context_lens, int block_size,
                          int max_context_len, int max_num_blocks_per_seq,
                          const float* alibi_slopes, int64_t q_stride,
                          int64_t kv_block_stride, int64_t kv_head_stride,
                          int act_dtype, int cache_dtype, void* stream);

/* No counterpart in the reference (whose caller allocates the three temporaries on every call,
 * vllm/model_executor/layers/attention/ops/paged_attn.py:83-93).  For callers that cannot
 * allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc: sw_paged_attention_v2 with exp_sums / max_logits / tmp_out taken from a
 * file-scope device array of sw_paged_attention_v2_static_max_bytes() = 32 MiB that every GPU the
 * library runs on holds until the process ends.  Allocates nothing, on the host side neither.
 * Rejected with SW_ERR_INVALID when
 *   2 * round256(4 * R) + round256(R * head_size * sizeof(act)) > 32 MiB,
 *   R = num_seqs * num_heads * ceil(max_context_len / 512)
 * (e.g. 256 sequences x 32 heads x 1024 tokens x 128 bf16: 4.3 MiB).  Results are bit-identical to
 * sw_paged_attention_v2's.  Every call on one GPU uses that GPU's one array: calls that overlap
 * on a GPU (two streams, two threads) race; order them on one stream.  Nothing in the array has
 * to be zero: each call writes every row it later reads.                                     */
int64_t sw_paged_attention_v2_static_max_bytes(void);
int sw_paged_attention_v2_static(void* out, const void* query, const void* key_cache,
                                 const void* value_cache, int num_seqs, int num_heads,
                                 int num_kv_heads, int head_size, float scale,
                                 const int32_t* block_tables,
                                 const int32_t* context_lens, int block_size,
                                 int max_context_len, int max_num_blocks_per_seq,
                                 const float* alibi_slopes, int64_t q_stride,
                                 int64_t kv_block_stride, int64_t kv_head_stride,
                                 int act_dtype, int cache_dtype, void* stream);

/* ---------------------------------------------------------- prefill attention
 * Causal self-attention of a prompt run.  Replaces the prompt branch of the
 * reference's attention backends, whose arithmetic is third-party:
 * xformers.ops.memory_efficient_attention_forward with BlockDiagonalCausalMask
 * (vllm/model_executor/layers/attention/backends/xformers.py:84-177) /
 * flash_attn_func (backends/flash_attn.py:89-103); semantics = _ref_masked_attention
 * (xformers.py:236-261).
 *   query [T, num_heads, head_size], key/value [T, num_kv_heads, head_size], out like
 *   query; row strides in elements.  Sequence i occupies rows
 *   [seq_starts[i], seq_starts[i] + seq_lens[i]) (padded or packed batches alike).
 *   alibi_slopes float [num_heads] or NULL; sliding_window <= 0 disables the window.
 * float16 / bfloat16 only; head_size in {64,80,96,112,128,256}.              */
int sw_prefill_attention(void* out, const void* query, const void* key,
                         const void* value, const int32_t* seq_starts,
                         const int32_t* seq_lens, int num_seqs, int max_seq_len,
                         int num_heads, int num_kv_heads, int head_size,
                         float scale, const float* alibi_slopes, int64_t q_stride,
                         int64_t k_stride, int64_t v_stride, int64_t out_stride,
                         int sliding_window, int act_dtype, void* stream);

/* --------------------------------------------------------------- KV transport
 * Replaces the reference's MSCCL++ proxy-channel kernels (csrc/kv_comm_kernels.cu:3-29) and
 * the memory registration of vllm/worker/comm_utils.py:171-207.
 *
 * sw_ipc_*: export a device allocation to / map it from another process on this node
 * (HIP IPC; peer GPUs are reached over xGMI).  A handle is 64 opaque bytes.            */
int sw_ipc_get_mem_handle(const void* dev_ptr, void* handle_out64);
/* Base address and size of the device allocation that contains dev_ptr.  A handle always names a
 * whole allocation: callers whose buffer is carved out of a larger one (a caching allocator's
 * segment) export the base and send the buffer's offset beside the handle, as
 * comm_utils.py:171-207 registers whole tensors by data_ptr and size.                        */
int sw_mem_address_range(const void* dev_ptr, void** base_out, int64_t* size_out);
int sw_ipc_open_mem_handle(const void* handle64, void** dev_ptr_out);
int sw_ipc_close_mem_handle(void* dev_ptr);

/* nw_cache_out_kernel / KVCacheCommunicator.put (comm_utils.py:132-147): copy num_runs byte
 * ranges src_base + offsets[i] -> dst_base + offsets[i] (same offset on both sides), each
 * nbytes[i] long; offsets and sizes are HOST arrays and multiples of 16.  dst_base is normally
 * the IPC mapping of the peer worker's cache.  Stream-ordered, non-blocking on the host.
 * max_workgroups > 0 caps the launch (each workgroup then walks several 64 KiB chunks): a push
 * over one xGMI link needs few workgroups to fill the link and should leave the compute units to
 * the kernels it runs beside; <= 0 launches one workgroup per chunk (device-local copies).     */
int sw_kv_push(void* dst_base, const void* src_base, const int64_t* offsets,
               const int64_t* nbytes, int num_runs, int max_workgroups, void* stream);

/* sw_kv_push with a destination offset of its own for every range: src_base + src_offsets[i] ->
 * dst_base + dst_offsets[i].  The reference registers the two caches under the same block numbers
 * (comm_utils.py:132-147 uses one offset for both sides); a prompt worker that keeps only the blocks
 * of the prompts in flight, or that feeds several token workers, numbers its blocks by itself.     */
int sw_kv_push_mapped(void* dst_base, const void* src_base, const int64_t* dst_offsets,
                      const int64_t* src_offsets, const int64_t* nbytes, int num_runs,
                      int max_workgroups, void* stream);

/* nw_cache_out_signal_kernel (kv_comm_kernels.cu:22-29): after a system-scope release,
 * flags[slot] := epoch in the PEER's flag array (IPC mapping).  Ordered after every earlier
 * sw_kv_push on the same stream.                                                         */
int sw_kv_signal(int32_t* flags, int slot, int32_t epoch, void* stream);

/* nw_cache_in_kernel (kv_comm_kernels.cu:3-9): wait on the stream until the LOCAL
 * flags[slot] has reached epoch, at most max_spins polls (the reference bounds its wait at
 * 1e8 spins); on a timeout *status (device or pinned-host word) := 1 + slot and the stream
 * continues.                                                                              */
int sw_kv_wait(const int32_t* flags, int slot, int32_t epoch, int64_t max_spins,
               int32_t* status, void* stream);

/* ------------------------------------------------ int4 weight-only linear layers
 * fp16 activations and scales only (vllm/model_executor/layers/quantization/gptq.py:50-51,
 * awq.py:45-46); 4-bit weights, and for the three sw_gptq_* entry points also 8-bit ones (qweight int32 [K/4, N],
 * byte i of a word = row 4r+i; qzeros int32 [groups, N/4]; csrc/quantization/gptq/qdq_8.cuh, q_gemm.cu:601-735,
 * :810-918, :1989-2029) through plain kernels without the matrix cores; the reference's 2/3-bit GPTQ kernels are
 * not provided: SW_ERR_INVALID.  W[k][n] = fp16((q - zero) * scale); C = A.W accumulated in fp32.
 *
 * Workspace: the fused GEMMs split k across workgroups and combine fp32 partial tiles in
 * device memory.  `workspace` must hold at least 16 KiB, whose first 16 KiB are ZERO the first
 * time it is passed and are left zero by every call; a larger workspace allows more splits
 * (sw_quant_gemm_workspace_bytes gives a size that never limits them).  Calls that share a
 * workspace must be ordered on one stream.                                              */
int64_t sw_quant_gemm_workspace_bytes(int M, int N, int K);

/* csrc/quantization/gptq/q_gemm.cu:2748-2769 gptq_shuffle: in place, once after loading.
 *   qweight  int32 [K/8, N], nibble i of a word = row 8r+i
 *   q_perm   int32 [K] or NULL: row order that makes groups sequential (argsort of g_idx);
 *            needs `scratch` of K/8*N*4 bytes (the reference allocates and frees its own)
 * Asynchronous on `stream` (the reference synchronises the device).                     */
int sw_gptq_shuffle(uint32_t* qweight, const int32_t* q_perm, int K, int N, int bit,
                    void* scratch, void* stream);

/* csrc/quantization/gptq/q_gemm.cu:2712-2746 gptq_gemm: c[M,N] = a[M,K] . W.
 *   qzeros int32 [groups, N/8] (nibble n%8, stored value is zero-1), scales fp16 [groups, N]
 *   use_exllama != 0: qweight was shuffled; g_idx = the q_perm given to sw_gptq_shuffle or NULL
 *   use_exllama == 0: plain qweight; g_idx int32 [K] = group of every row (required)
 * One code path for every M (the reference switches to reconstruct + BLAS above 50 rows).  */
int sw_gptq_gemm(void* c, const void* a, const uint32_t* qweight, const uint32_t* qzeros,
                 const void* scales, const int32_t* g_idx, int M, int N, int K, int groups,
                 int use_exllama, int bit, void* workspace, int64_t workspace_bytes,
                 void* stream);

/* csrc/quantization/gptq/q_gemm.cu:2820-2861 dequant_gptq (kernels :920-1037, :1517-1561):
 * out fp16 [K, N] in ORIGINAL row order, bit-identical to the reference's reconstruction.  */
int sw_gptq_dequant(void* out, const uint32_t* qweight, const uint32_t* qzeros,
                    const void* scales, const int32_t* g_idx, int K, int N, int groups,
                    int use_exllama, int bit, void* stream);

/* csrc/quantization/awq/gemm_kernels.cu:648-705 awq_gemm: c[M,N] = a[M,K] . W.
 *   qweight int32 [K, N/8], qzeros int32 [K/group_size, N/8], scales fp16 [K/group_size, N];
 *   nibble j of a word is logical column {0,2,4,6,1,3,5,7}[j] of its 8 columns
 *   (dequantize.cuh:40-56).  The reference's split_k_iters has no counterpart: the split
 *   is chosen here and the partials are combined in the same launch.
 * Errors mirror the reference's argument checks (:671-678).                               */
int sw_awq_gemm(void* c, const void* a, const uint32_t* qweight, const void* scales,
                const uint32_t* qzeros, int M, int K, int N, int group_size,
                void* workspace, int64_t workspace_bytes, void* stream);

/* csrc/quantization/awq/gemm_kernels.cu:585-640 awq_dequantize: out fp16 [K, N].          */
int sw_awq_dequantize(void* out, const uint32_t* qweight, const void* scales,
                      const uint32_t* qzeros, int K, int N, int group_size, void* stream);

/* No counterpart in the reference, which keeps two sets of int4 kernels: an AWQ checkpoint
 * (csrc/quantization/awq/dequantize.cuh:17-84: a word = 8 columns of one row, columns in the
 * order 0 2 4 6 1 3 5 7, zeros stored as they are) repacked ONCE into the layout of the GPTQ
 * kernels after gptq_shuffle (q_gemm.cu:2075, qdq_4.cuh:12-14: a word = 8 consecutive rows of
 * one column), so that one set of kernels serves both formats.  Out of place:
 *   qweight_out int32 [K/8, N], qzeros_out int32 [groups, N/8] in plain column order.
 * The reconstructed fp16 weights are bit-identical to sw_awq_dequantize's.                 */
int sw_awq_to_kmajor(uint32_t* qweight_out, uint32_t* qzeros_out, const uint32_t* qweight,
                     const uint32_t* qzeros, int K, int N, int groups, void* stream);

/* sw_gptq_gemm / sw_gptq_dequant on shuffled weights without act-order, with the value added
 * to the stored zero as an argument: 1 = GPTQ checkpoints (q_gemm.cu:929 `zeros[i] + 1`),
 * 0 = weights from sw_awq_to_kmajor (dequantize.cuh:60-75 subtracts the zero as stored).   */
int sw_int4_kmajor_gemm(void* c, const void* a, const uint32_t* qweight, const uint32_t* qzeros,
                        const void* scales, int M, int N, int K, int groups, int zero_offset,
                        void* workspace, int64_t workspace_bytes, void* stream);
int sw_int4_kmajor_dequant(void* out, const uint32_t* qweight, const uint32_t* qzeros,
                           const void* scales, int K, int N, int groups, int zero_offset,
                           void* stream);

/* -------------------------------------------- row-wise operators beside the path
 * csrc/layernorm_kernels.cu:70-118 rms_norm / fused_add_rms_norm.
 *   out = (act)(x * rsqrt(mean(x^2) + eps)) * weight, per row of hidden_size elements.
 *   fused: residual := input + residual (stored), input := norm(residual) * weight.
 *   input rows may be strided (input_stride elements); out / residual are contiguous. */
int sw_rms_norm(void* out, const void* input, const void* weight, float epsilon,
                int64_t num_tokens, int hidden_size, int64_t input_stride,
                int act_dtype, void* stream);
int sw_fused_add_rms_norm(void* input, void* residual, const void* weight,
                          float epsilon, int64_t num_tokens, int hidden_size,
                          int64_t input_stride, int act_dtype, void* stream);

/* csrc/pos_encoding_kernels.cu:120-171 rotary_embedding, in place on query and key.
 *   positions int64 [num_tokens]; cos_sin_cache act_dtype [max_position, rot_dim]
 *   (cos in the first half of a row, sin in the second); is_neox selects the
 *   (i, i + rot_dim/2) pairing, otherwise (2i, 2i+1).  Products are formed in fp32
 *   and rounded once (the reference rounds each half-precision operation).   */
int sw_rotary_embedding(const int64_t* positions, void* query, void* key,
                        int64_t num_tokens, int num_heads, int num_kv_heads,
                        int head_size, const void* cos_sin_cache, int rot_dim,
                        int is_neox, int64_t query_stride, int64_t key_stride,
                        int act_dtype, void* stream);

/* rotary_embedding (csrc/pos_encoding_kernels.cu:120-171) and reshape_and_cache
 * (csrc/cache_kernels.cu:228-268) of the same tokens in one launch, for decode steps (one token
 * per sequence: vllm/model_executor/models/llama.py:150-156 calls the first, layers/attention/
 * ops/paged_attn.py:56-71 the second).  query and key are rotated in place, the rotated key and
 * the value go to the caches at slot_mapping[t] (negative: not cached).  Arguments as in the two
 * entry points above; results identical to calling them one after the other, bit for bit.
 * head_size, rot_dim/2 (neox) or rot_dim (gpt-j) and head_size - rot_dim must span whole 16-byte
 * vectors: SW_ERR_INVALID otherwise (callers then make the two calls).                      */
int sw_rotary_embedding_and_cache(const int64_t* positions, void* query, void* key,
                                  const void* value, void* key_cache, void* value_cache,
                                  const int64_t* slot_mapping, int64_t num_tokens, int num_heads,
                                  int num_kv_heads, int head_size, const void* cos_sin_cache,
                                  int rot_dim, int is_neox, int block_size, int x,
                                  int64_t query_stride, int64_t key_stride, int64_t value_stride,
                                  int act_dtype, int cache_dtype, void* stream);

/* csrc/activation_kernels.cu:66-84 silu_and_mul: out[t, i] = silu(in[t, i]) * in[t, d + i],
 * input [num_tokens, 2 * d] with row stride input_stride, out [num_tokens, d].       */
int sw_silu_and_mul(void* out, const void* input, int64_t num_tokens, int d,
                    int64_t input_stride, int act_dtype, void* stream);

/* ------------------------------------------------- prefill attention over a cached prefix
 * Replaces the Triton kernels behind context_attention_fwd
 * (vllm/model_executor/layers/attention/ops/prefix_prefill.py:10-745; caller
 * PagedAttentionImpl.forward_prefix, ops/paged_attn.py:113-138): sequence i has ctx_lens[i]
 * tokens already in the paged cache (block_tables row i) and new_lens[i] new tokens in rows
 * [seq_starts[i], +new_lens[i]) of query / key / value; new token j attends to the whole
 * context and to new tokens 0..j.  key_cache [blocks, num_kv_heads, head_size/8, block_size, 8],
 * value_cache [blocks, num_kv_heads, head_size, block_size], 16-bit elements of the activation
 * dtype, strides in elements; block_size a multiple of 4.  Other arguments as in
 * sw_prefill_attention.                                                                      */
int sw_context_attention(void* out, const void* query, const void* key, const void* value,
                         const void* key_cache, const void* value_cache, const int32_t* block_tables,
                         int max_blocks_per_seq, const int32_t* seq_starts, const int32_t* new_lens,
                         const int32_t* ctx_lens, int num_seqs, int max_new_len, int num_heads,
                         int num_kv_heads, int head_size, int block_size, float scale,
                         const float* alibi_slopes, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                         int64_t out_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                         int sliding_window, int act_dtype, void* stream);

/* sw_context_attention with the key stages of every (query tile, sequence, head) -- the cached context, then the
 * new tokens -- dealt out to num_parts workgroups, whose partial results (fp32, in `workspace`) a second kernel
 * joins.  For launches that leave compute units idle otherwise: few new tokens over a long cached context (the
 * chunks of a chunked prefill, a short question after a long shared prefix).  The reference's Triton kernel walks
 * the whole context in one program per (sequence, head, 128-row block) (ops/prefix_prefill.py:621-745).
 *   total_rows   rows of query / key / value / out (the partial results are indexed by row)
 *   num_parts    1 = sw_context_attention; sw_context_attention_split_parts suggests a value from the launch's
 *                shape and the longest possible context (max_context_len, e.g. max_blocks_per_seq * block_size)
 *   workspace    at least sw_context_attention_split_workspace_bytes(...) bytes, 16-byte aligned; contents
 *                need not be initialised; calls that share it must be ordered on one stream
 * Head sizes that are multiples of 32 and block sizes that are multiples of 8 only (SW_ERR_INVALID otherwise;
 * sw_context_attention_split_parts returns 1 for the others).  Results agree with sw_context_attention's within
 * the rounding of one more fp32 summation.                                                             */
int64_t sw_context_attention_split_workspace_bytes(int num_parts, int64_t total_rows, int num_heads,
                                                   int head_size);
int sw_context_attention_split_parts(int num_seqs, int max_new_len, int num_heads, int head_size,
                                     int block_size, int max_context_len);
int sw_context_attention_split(void* out, const void* query, const void* key, const void* value,
                               const void* key_cache, const void* value_cache, const int32_t* block_tables,
                               int max_blocks_per_seq, const int32_t* seq_starts, const int32_t* new_lens,
                               const int32_t* ctx_lens, int num_seqs, int max_new_len, int num_heads,
                               int num_kv_heads, int head_size, int block_size, float scale,
                               const float* alibi_slopes, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                               int64_t out_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                               int sliding_window, int act_dtype, int64_t total_rows, int num_parts,
                               void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------- all-reduce by direct peer reads
 * Replaces csrc/custom_all_reduce.cu:12-148 / custom_all_reduce.cuh:341-548 (init_custom_ar,
 * all_reduce_reg / all_reduce_unreg, dispose, meta_size), which the reference builds for CUDA only.
 * Every rank owns ONE device allocation that its peers map (sw_ipc_*): at signal_off a signal area
 * of sw_custom_ar_signal_bytes() bytes, zeroed before any rank's first call; at scratch_off and at
 * staging_off max_bytes bytes each.  peer_bases[r] is rank r's allocation as mapped in this
 * process (the local pointer for r == rank).  Calls must be made in the same order with the same
 * sizes on every rank; one call at a time per state.                                          */
int64_t sw_custom_ar_signal_bytes(void);
int sw_custom_ar_init(void** peer_bases, int rank, int world, int64_t signal_off, int64_t scratch_off,
                      int64_t staging_off, int64_t max_bytes, void** state_out);
/* output = sum over ranks of their inputs (fp32 accumulation in the reference's rank order, one
 * rounding; bitwise the same on every rank).  input: nbytes device bytes, copied to the staging
 * area on the stream first; NULL (or the staging area itself) when the caller has produced the
 * data there already (the reference's registered-buffer case).  act_dtype SW_F32 / SW_F16 /
 * SW_BF16.  algorithm 0 = the reference's size rule (custom_all_reduce.cuh:521-527), 1 = one
 * stage, 2 = reduce-scatter + all-gather.                                                     */
int sw_custom_all_reduce(void* state, const void* input, void* output, int64_t nbytes, int act_dtype,
                         int algorithm, void* stream);
int sw_custom_ar_dispose(void* state);

#ifdef __cplusplus
}
#endif
#endif /* SPLITWISE_HIP_H_ */
