// C-ABI entry points for decode paged attention (argument checks + dtype dispatch).
// The kernels live in paged_attention.cuh and are instantiated in pa_*.hip.
#include "paged_attention.cuh"

#include <mutex>

namespace sw {

// Temporaries of the partitioned kernel for callers that cannot allocate (sw_paged_attention_v2_static):
// one array per GPU, held until the process ends.  In host code the array's NAME is only a handle for
// hipGetSymbolAddress; the device address it yields is what kernels and copies are given.
constexpr int64_t kStaticScratchBytes = int64_t{32} << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
static void* g_scratch_ptr[kMaxDevices];          // device address of `scratch` on each GPU, once looked up
static std::mutex g_scratch_mutex;

// Device address of this GPU's copy of `scratch` (looked up once per device, with that device current).
static int static_scratch(unsigned char** out) {
  int dev = -1;
  SW_CHECK_HIP(hipGetDevice(&dev));
  SW_CHECK_ARG(dev >= 0 && dev < kMaxDevices, "static scratch: device %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void* ptr = nullptr;
    SW_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    SW_CHECK_ARG(ptr != nullptr, "static scratch: hipGetSymbolAddress returned a null address");
    g_scratch_ptr[dev] = ptr;
  }
  *out = static_cast<unsigned char*>(g_scratch_ptr[dev]);
  return SW_OK;
}

static int run_paged_attention(PaArgs& a, int act_dtype, int cache_dtype) {
  if (a.num_seqs == 0) return SW_OK;
  SW_CHECK_ARG(a.num_seqs > 0 && a.num_heads > 0 && a.num_kv_heads > 0, "paged_attention: bad shape");
  SW_CHECK_ARG(a.num_heads % a.num_kv_heads == 0,
               "paged_attention: num_heads %d not divisible by num_kv_heads %d", a.num_heads,
               a.num_kv_heads);
  SW_CHECK_ARG(a.num_seqs <= 65535, "paged_attention: more than 65535 sequences in one call");
  SW_CHECK_ARG(act_dtype == SW_F32 || act_dtype == SW_F16 || act_dtype == SW_BF16,
               "Unsupported data type: %d", act_dtype);
  const bool fp8 = cache_dtype == SW_FP8_E5M2;
  SW_CHECK_ARG(fp8 || cache_dtype == act_dtype, "Unsupported data type of kv cache: %d", cache_dtype);
  const int cs = fp8 ? 1 : dtype_size(act_dtype);
  SW_CHECK_ARG((a.kv_block_stride * cs) % 16 == 0 && (a.kv_head_stride * cs) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(a.k) % 16 == 0 && reinterpret_cast<uintptr_t>(a.v) % 16 == 0,
               "paged_attention: cache strides / base must be 16-byte aligned");
  // 16-bit queries are read as (even, odd) pairs
  SW_CHECK_ARG(dtype_size(act_dtype) != 2 || (a.q_stride % 2 == 0 && reinterpret_cast<uintptr_t>(a.q) % 4 == 0),
               "paged_attention: query rows must be 4-byte aligned");
  SW_CHECK_ARG(a.max_context_len > 0 && a.max_num_blocks_per_seq > 0, "paged_attention: empty context");
  if (a.partitioned) {
    SW_CHECK_ARG(a.exp_sums && a.max_logits && a.tmp_out, "paged_attention_v2: missing temporaries");
    SW_CHECK_ARG((a.max_context_len + kPartitionSize - 1) / kPartitionSize <= 65535,
                 "paged_attention_v2: too many partitions");
  }
  switch (act_dtype) {
    case SW_F16: return fp8 ? pa_dispatch_f16_fp8(a) : pa_dispatch_f16(a);
    case SW_BF16: return fp8 ? pa_dispatch_bf16_fp8(a) : pa_dispatch_bf16(a);
    default: return fp8 ? pa_dispatch_f32_fp8(a) : pa_dispatch_f32(a);
  }
}

}  // namespace sw

extern "C" {

int sw_paged_attention_v1(void* out, const void* query, const void* key_cache,
                          const void* value_cache, int num_seqs, int num_heads, int num_kv_heads,
                          int head_size, float scale, const int32_t* block_tables,
                          const int32_t* context_lens, int block_size, int max_context_len,
                          int max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
                          int64_t kv_block_stride, int64_t kv_head_stride, int act_dtype,
                          int cache_dtype, void* stream) {
  sw::PaArgs a{out, nullptr, nullptr, nullptr, query, key_cache, value_cache, num_seqs, num_heads,
               num_kv_heads, head_size, scale, block_tables, context_lens, block_size,
               max_context_len, max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
               kv_head_stride, 0, static_cast<hipStream_t>(stream)};
  return sw::run_paged_attention(a, act_dtype, cache_dtype);
}

int sw_paged_attention_v2(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                          const void* query, const void* key_cache, const void* value_cache,
                          int num_seqs, int num_heads, int num_kv_heads, int head_size, float scale,
                          const int32_t* block_tables, const int32_t* context_lens, int block_size,
                          int max_context_len, int max_num_blocks_per_seq,
                          const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                          int64_t kv_head_stride, int act_dtype, int cache_dtype, void* stream) {
  sw::PaArgs a{out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs,
               num_heads, num_kv_heads, head_size, scale, block_tables, context_lens, block_size,
               max_context_len, max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
               kv_head_stride, 1, static_cast<hipStream_t>(stream)};
  return sw::run_paged_attention(a, act_dtype, cache_dtype);
}

int64_t sw_paged_attention_v2_static_max_bytes(void) { return sw::kStaticScratchBytes; }

int sw_paged_attention_v2_static(void* out, const void* query, const void* key_cache,
                                 const void* value_cache, int num_seqs, int num_heads, int num_kv_heads,
                                 int head_size, float scale, const int32_t* block_tables,
                                 const int32_t* context_lens, int block_size, int max_context_len,
                                 int max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
                                 int64_t kv_block_stride, int64_t kv_head_stride, int act_dtype,
                                 int cache_dtype, void* stream) {
  if (num_seqs == 0) return SW_OK;
  SW_CHECK_ARG(num_seqs > 0 && num_heads > 0 && head_size > 0 && max_context_len > 0,
               "paged_attention_v2_static: bad shape");
  SW_CHECK_ARG(sw::dtype_size(act_dtype) > 0 && act_dtype != SW_FP8_E5M2, "Unsupported data type: %d", act_dtype);
  const int64_t parts = (static_cast<int64_t>(max_context_len) + sw::kPartitionSize - 1) / sw::kPartitionSize;
  const int64_t rows = static_cast<int64_t>(num_seqs) * num_heads * parts;
  auto round256 = [](int64_t n) { return (n + 255) / 256 * 256; };
  const int64_t floats_bytes = round256(rows * 4);
  const int64_t tmp_bytes = round256(rows * head_size * sw::dtype_size(act_dtype));
  const int64_t need = 2 * floats_bytes + tmp_bytes;
  SW_CHECK_ARG(need <= sw::kStaticScratchBytes,
               "paged_attention_v2_static: %d sequences x %d heads x %lld partitions x head size %d need %lld "
               "bytes of temporaries, the static scratch holds %lld; use sw_paged_attention_v2",
               num_seqs, num_heads, static_cast<long long>(parts), head_size, static_cast<long long>(need),
               static_cast<long long>(sw::kStaticScratchBytes));
  unsigned char* base = nullptr;
  if (int rc = sw::static_scratch(&base)) return rc;
  // every partition row that the reduce kernel reads is written by the partition kernel of the same
  // call, so nothing in the scratch has to start at zero
  float* exp_sums = reinterpret_cast<float*>(base);
  float* max_logits = reinterpret_cast<float*>(base + floats_bytes);
  void* tmp_out = base + 2 * floats_bytes;
  sw::PaArgs a{out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs,
               num_heads, num_kv_heads, head_size, scale, block_tables, context_lens, block_size,
               max_context_len, max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
               kv_head_stride, 1, static_cast<hipStream_t>(stream)};
  return sw::run_paged_attention(a, act_dtype, cache_dtype);
}

}  // extern "C"
