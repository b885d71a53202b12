// KV-cache maintenance kernels for gfx950: reshape_and_cache, copy_blocks,
// swap_blocks, convert_fp8_e5m2.
//
// Semantics follow csrc/cache_kernels.cu of the reference (cited per entry
// point in include/splitwise_hip.h); the tiling does not.  All four are HBM-bound
// byte movers, so the design goal is 16-byte accesses that land in whole 128-byte
// lines:
//   * reshape_and_cache works on tiles of 16 tokens x one head staged through
//     LDS, so that the scatter into the paged layout is issued token-fastest.
//     With consecutive slots (a prompt) one wave store covers contiguous
//     256-byte (K) / 1-KiB (V) pieces of the 4-KiB [block, head] cache tile; with
//     arbitrary slots (decode) every token is still placed individually.
//   * copy_blocks moves 16 bytes per lane with the pointer tables resident on
//     the device (no per-call pointer upload).
#include "common.cuh"

#include <algorithm>
#include <string>
#include <vector>

namespace sw {

namespace {
thread_local std::string g_last_error;
}

void set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

constexpr int kTileTokens = 16;
constexpr int kRcThreads = 256;

// LDS row = head_size activations + 16 bytes of padding (keeps 16-byte row reads
// of consecutive tokens on different bank groups).
template <typename act_t>
__device__ __forceinline__ int lds_row_bytes(int head_size) {
  return head_size * (int)sizeof(typename Conv<act_t>::store_t) + 16;
}

// act_t: float / f16_t / bf16_t.  FP8: cache holds e5m2 bytes, else the same
// bits as the activations.
template <typename act_t, bool FP8>
__global__ void __launch_bounds__(kRcThreads)
reshape_and_cache_tile_kernel(const char* __restrict__ key, const char* __restrict__ value,
                              char* __restrict__ key_cache, char* __restrict__ value_cache,
                              const int64_t* __restrict__ slot_mapping, int num_tokens,
                              int num_heads, int head_size, int block_size,
                              int64_t key_stride_bytes, int64_t value_stride_bytes) {
  using S = typename Conv<act_t>::store_t;
  constexpr int ES = sizeof(S);                 // activation element bytes
  constexpr int CS = FP8 ? 1 : ES;              // cache element bytes
  constexpr int X = 16 / CS;                    // cache elements per 16 bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t0 = blockIdx.x * kTileTokens;
  const int head = blockIdx.y;
  const int tid = threadIdx.x;
  const int row_bytes = lds_row_bytes<act_t>(head_size);
  char* k_tile = smem;
  char* v_tile = smem + kTileTokens * row_bytes;
  int64_t* slots = reinterpret_cast<int64_t*>(smem + 2 * kTileTokens * row_bytes);

  if (tid < kTileTokens) {
    const int t = t0 + tid;
    slots[tid] = (t < num_tokens) ? slot_mapping[t] : -1;
  }

  // ---- stage K and V rows of this head: 16-byte pieces, token-major (coalesced)
  const int chunks_per_row = head_size * ES / 16;
  const int n_in = kTileTokens * chunks_per_row;
  for (int i = tid; i < n_in; i += kRcThreads) {
    const int tok = i / chunks_per_row;
    const int ch = i - tok * chunks_per_row;
    const int t = t0 + tok;
    if (t < num_tokens) {
      const int64_t off = (int64_t)head * head_size * ES + (int64_t)ch * 16;
      const uint4 kv = *reinterpret_cast<const uint4*>(key + t * key_stride_bytes + off);
      const uint4 vv = *reinterpret_cast<const uint4*>(value + t * value_stride_bytes + off);
      *reinterpret_cast<uint4*>(k_tile + tok * row_bytes + ch * 16) = kv;
      *reinterpret_cast<uint4*>(v_tile + tok * row_bytes + ch * 16) = vv;
    }
  }
  __syncthreads();

  // ---- K: one 16-byte cache vector (X elements of one token) per lane, token-fastest
  const int k_vecs = head_size / X;
  const int64_t k_head_bytes = (int64_t)head_size * block_size * CS;
  const int64_t k_block_bytes = k_head_bytes * num_heads;
  for (int i = tid; i < k_vecs * kTileTokens; i += kRcThreads) {
    const int tok = i % kTileTokens;
    const int c = i / kTileTokens;
    const int64_t slot = slots[tok];
    if (slot < 0) continue;
    const int64_t blk = slot / block_size;
    const int boff = (int)(slot - blk * block_size);
    char* dst = key_cache + blk * k_block_bytes + head * k_head_bytes +
                ((int64_t)c * block_size + boff) * 16;
    const char* src = k_tile + tok * row_bytes + c * X * ES;
    if constexpr (!FP8) {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
    } else {
      uint32_t w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t acc = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const S e = *reinterpret_cast<const S*>(src + (q * 4 + b) * ES);
          acc |= (uint32_t)Conv<act_t>::to_e5m2(e) << (8 * b);
        }
        w[q] = acc;
      }
      *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }

  // ---- V: cache row d holds block_size tokens.  Fast path when the tile's
  // live tokens occupy consecutive slots of ONE block starting at a multiple of
  // VT: 16-byte stores of VT tokens.  Otherwise one element per lane.
  constexpr int VT_MAX = 16 / CS;
  const int64_t v_head_bytes = k_head_bytes;
  const int64_t v_block_bytes = k_block_bytes;
  const int vt = min(VT_MAX, block_size);       // tokens per vector store (4, 8 or 16)
  const int parts = kTileTokens / vt;           // token groups in this tile
  const int vbytes = vt * CS;                   // 16, or 8 for fp8 with block_size 8
  // A group of vt tokens takes the vector path when its slots are consecutive
  // and start at a multiple of vt (then they share one cache row segment).
  for (int i = tid; i < head_size * parts; i += kRcThreads) {
    const int part = i % parts;
    const int d = i / parts;
    const int64_t g0 = slots[part * vt];
    bool fast = g0 >= 0 && (g0 % vt) == 0;
    for (int j = 1; j < vt; ++j) fast = fast && (slots[part * vt + j] == g0 + j);
    if (fast) {
      const int64_t blk = g0 / block_size;
      const int boff = (int)(g0 - blk * block_size);
      char* dst = value_cache + blk * v_block_bytes + head * v_head_bytes +
                  ((int64_t)d * block_size + boff) * CS;
      uint32_t w[4] = {0, 0, 0, 0};
      for (int j = 0; j < vt; ++j) {
        const S e = *reinterpret_cast<const S*>(v_tile + (part * vt + j) * row_bytes + d * ES);
        if constexpr (FP8) {
          w[j >> 2] |= (uint32_t)Conv<act_t>::to_e5m2(e) << (8 * (j & 3));
        } else if constexpr (ES == 2) {
          w[j >> 1] |= (uint32_t)e << (16 * (j & 1));
        } else {
          w[j] = __float_as_uint(e);
        }
      }
      if (vbytes == 16) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
      } else {
        *reinterpret_cast<uint2*>(dst) = make_uint2(w[0], w[1]);
      }
    } else {
      for (int j = 0; j < vt; ++j) {
        const int tok = part * vt + j;
        const int64_t slot = slots[tok];
        if (slot < 0) continue;
        const int64_t blk = slot / block_size;
        const int boff = (int)(slot - blk * block_size);
        char* dst = value_cache + blk * v_block_bytes + head * v_head_bytes +
                    ((int64_t)d * block_size + boff) * CS;
        const S e = *reinterpret_cast<const S*>(v_tile + tok * row_bytes + d * ES);
        if constexpr (FP8) {
          *reinterpret_cast<uint8_t*>(dst) = Conv<act_t>::to_e5m2(e);
        } else {
          *reinterpret_cast<S*>(dst) = e;
        }
      }
    }
  }
}

// Grid: (2 * num_pairs, num_layers, splits).  blockIdx.x even = K, odd = V.
__global__ void __launch_bounds__(256)
copy_blocks_kernel(const void* const* __restrict__ key_ptrs,
                   const void* const* __restrict__ value_ptrs,
                   const int64_t* __restrict__ mapping, int64_t vecs_per_block) {
  const int pair = blockIdx.x >> 1;
  const bool is_v = blockIdx.x & 1;
  const int layer = blockIdx.y;
  uint4* base = reinterpret_cast<uint4*>(const_cast<void*>(is_v ? value_ptrs[layer] : key_ptrs[layer]));
  const int64_t src = mapping[2 * pair];
  const int64_t dst = mapping[2 * pair + 1];
  const uint4* s = base + src * vecs_per_block;
  uint4* d = base + dst * vecs_per_block;
  for (int64_t i = (int64_t)blockIdx.z * blockDim.x + threadIdx.x; i < vecs_per_block;
       i += (int64_t)gridDim.z * blockDim.x) {
    d[i] = s[i];
  }
}

// 8 elements per lane per step: 16 bytes of f16/bf16 (32 of f32) <-> 8 bytes of e5m2.
template <typename act_t, bool TO_FP8>
__global__ void __launch_bounds__(256)
convert_fp8_e5m2_kernel(const char* __restrict__ src, char* __restrict__ dst, int64_t numel) {
  using S = typename Conv<act_t>::store_t;
  const int64_t nvec = numel / 8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    if constexpr (TO_FP8) {
      S e[8];
      if constexpr (sizeof(S) == 2) {
        *reinterpret_cast<uint4*>(e) = reinterpret_cast<const uint4*>(src)[i];
      } else {
        reinterpret_cast<uint4*>(e)[0] = reinterpret_cast<const uint4*>(src)[2 * i];
        reinterpret_cast<uint4*>(e)[1] = reinterpret_cast<const uint4*>(src)[2 * i + 1];
      }
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo |= (uint32_t)Conv<act_t>::to_e5m2(e[j]) << (8 * j);
        hi |= (uint32_t)Conv<act_t>::to_e5m2(e[4 + j]) << (8 * j);
      }
      reinterpret_cast<uint2*>(dst)[i] = make_uint2(lo, hi);
    } else {
      const uint2 q = reinterpret_cast<const uint2*>(src)[i];
      S e[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        e[j] = Conv<act_t>::from_e5m2((q.x >> (8 * j)) & 0xff);
        e[4 + j] = Conv<act_t>::from_e5m2((q.y >> (8 * j)) & 0xff);
      }
      if constexpr (sizeof(S) == 2) {
        reinterpret_cast<uint4*>(dst)[i] = *reinterpret_cast<uint4*>(e);
      } else {
        reinterpret_cast<uint4*>(dst)[2 * i] = reinterpret_cast<uint4*>(e)[0];
        reinterpret_cast<uint4*>(dst)[2 * i + 1] = reinterpret_cast<uint4*>(e)[1];
      }
    }
  }
  // tail (numel % 8 elements), handled by the first lanes of block 0
  const int64_t tail0 = nvec * 8;
  if (blockIdx.x == 0) {
    for (int64_t i = tail0 + threadIdx.x; i < numel; i += blockDim.x) {
      if constexpr (TO_FP8) {
        reinterpret_cast<uint8_t*>(dst)[i] = Conv<act_t>::to_e5m2(reinterpret_cast<const S*>(src)[i]);
      } else {
        reinterpret_cast<S*>(dst)[i] = Conv<act_t>::from_e5m2(reinterpret_cast<const uint8_t*>(src)[i]);
      }
    }
  }
}

template <typename act_t, bool FP8>
int launch_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                             const int64_t* slot_mapping, int num_tokens, int num_heads,
                             int head_size, int block_size, int64_t key_stride, int64_t value_stride,
                             hipStream_t stream) {
  using S = typename Conv<act_t>::store_t;
  const int es = sizeof(S);
  const int row_bytes = head_size * es + 16;
  const size_t smem = 2 * kTileTokens * row_bytes + kTileTokens * sizeof(int64_t);
  dim3 grid((num_tokens + kTileTokens - 1) / kTileTokens, num_heads);
  reshape_and_cache_tile_kernel<act_t, FP8><<<grid, kRcThreads, smem, stream>>>(
      static_cast<const char*>(key), static_cast<const char*>(value), static_cast<char*>(key_cache),
      static_cast<char*>(value_cache), slot_mapping, num_tokens, num_heads, head_size, block_size,
      key_stride * es, value_stride * es);
  SW_CHECK_LAUNCH();
  return SW_OK;
}

}  // namespace sw

extern "C" {

const char* sw_last_error(void) { return sw::g_last_error.c_str(); }

int sw_abi_version(void) { return 7; }       // SW_ABI_VERSION of the header; tests/test_abi.py compares the two

int sw_get_max_shared_memory_per_block(int device, int* out_bytes) {
  SW_CHECK_ARG(out_bytes != nullptr, "out_bytes is NULL");
  int v = 0;
  SW_CHECK_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
  *out_bytes = v;
  return SW_OK;
}

int sw_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                         const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
                         int block_size, int x, int64_t key_stride, int64_t value_stride,
                         int act_dtype, int cache_dtype, void* stream) {
  using namespace sw;
  if (num_tokens == 0) return SW_OK;
  SW_CHECK_ARG(num_tokens > 0 && num_heads > 0 && head_size > 0 && block_size > 0,
               "reshape_and_cache: bad shape");
  const bool fp8 = cache_dtype == SW_FP8_E5M2;
  SW_CHECK_ARG(act_dtype == SW_F32 || act_dtype == SW_F16 || act_dtype == SW_BF16,
               "Unsupported data type of key/value: %d", act_dtype);
  SW_CHECK_ARG(fp8 || cache_dtype == act_dtype, "Unsupported data type of kv cache: %d", cache_dtype);
  const int es = dtype_size(act_dtype);
  const int cs = fp8 ? 1 : es;
  SW_CHECK_ARG(x == 16 / cs, "reshape_and_cache: x=%d but 16/sizeof(cache element)=%d", x, 16 / cs);
  SW_CHECK_ARG(head_size % x == 0 && (head_size * es) % 16 == 0,
               "reshape_and_cache: head_size %d is not a multiple of x=%d", head_size, x);
  SW_CHECK_ARG((key_stride * es) % 16 == 0 && (value_stride * es) % 16 == 0 &&
                   (reinterpret_cast<uintptr_t>(key) % 16) == 0 &&
                   (reinterpret_cast<uintptr_t>(value) % 16) == 0,
               "reshape_and_cache: key/value rows must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
#define SW_RC(ACT)                                                                              \
  return fp8 ? launch_reshape_and_cache<ACT, true>(key, value, key_cache, value_cache,          \
                                                   slot_mapping, num_tokens, num_heads,         \
                                                   head_size, block_size, key_stride,           \
                                                   value_stride, s)                             \
             : launch_reshape_and_cache<ACT, false>(key, value, key_cache, value_cache,         \
                                                    slot_mapping, num_tokens, num_heads,        \
                                                    head_size, block_size, key_stride,          \
                                                    value_stride, s)
  switch (act_dtype) {
    case SW_F32: SW_RC(float);
    case SW_F16: SW_RC(f16_t);
    default: SW_RC(bf16_t);
  }
#undef SW_RC
}

int sw_copy_blocks(const void* const* key_cache_ptrs, const void* const* value_cache_ptrs,
                   int num_layers, const int64_t* block_mapping, int num_pairs,
                   int64_t block_bytes, void* stream) {
  using namespace sw;
  if (num_layers == 0 || num_pairs == 0) return SW_OK;
  SW_CHECK_ARG(num_layers > 0 && num_pairs > 0 && block_bytes > 0 && block_bytes % 16 == 0,
               "copy_blocks: bad arguments (layers=%d pairs=%d block_bytes=%lld)", num_layers,
               num_pairs, (long long)block_bytes);
  SW_CHECK_ARG(num_layers <= 65535, "copy_blocks: too many layers");
  const int64_t vecs = block_bytes / 16;
  const int splits = (int)std::min<int64_t>(32, std::max<int64_t>(1, vecs / 1024));
  dim3 grid(2 * num_pairs, num_layers, splits);
  copy_blocks_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(
      key_cache_ptrs, value_cache_ptrs, block_mapping, vecs);
  SW_CHECK_LAUNCH();
  return SW_OK;
}

int sw_swap_blocks(const void* src, void* dst, const int64_t* src_blocks, const int64_t* dst_blocks,
                   int n, int64_t block_bytes, int kind, void* stream) {
  using namespace sw;
  if (n == 0) return SW_OK;
  SW_CHECK_ARG(n > 0 && block_bytes > 0, "swap_blocks: bad arguments");
  hipMemcpyKind k;
  switch (kind) {
    case 0: k = hipMemcpyDeviceToDevice; break;
    case 1: k = hipMemcpyDeviceToHost; break;
    case 2: k = hipMemcpyHostToDevice; break;
    default: SW_CHECK_ARG(false, "Invalid device combination");
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  int i = 0;
  while (i < n) {
    int j = i + 1;
    while (j < n && src_blocks[j] == src_blocks[j - 1] + 1 && dst_blocks[j] == dst_blocks[j - 1] + 1) ++j;
    SW_CHECK_HIP(hipMemcpyAsync(static_cast<char*>(dst) + dst_blocks[i] * block_bytes,
                                static_cast<const char*>(src) + src_blocks[i] * block_bytes,
                                (size_t)(j - i) * block_bytes, k, s));
    i = j;
  }
  return SW_OK;
}

int sw_convert_fp8_e5m2(const void* src, void* dst, int64_t numel, int src_dtype, int dst_dtype,
                        void* stream) {
  using namespace sw;
  if (numel == 0) return SW_OK;
  const bool to_fp8 = dst_dtype == SW_FP8_E5M2;
  const int act = to_fp8 ? src_dtype : dst_dtype;
  SW_CHECK_ARG((src_dtype == SW_FP8_E5M2) != (dst_dtype == SW_FP8_E5M2),
               "convert_fp8_e5m2: exactly one side must be fp8_e5m2");
  SW_CHECK_ARG(act == SW_F32 || act == SW_F16 || act == SW_BF16, "convert_fp8_e5m2: bad dtype %d", act);
  SW_CHECK_ARG(reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0,
               "convert_fp8_e5m2: buffers must be 16-byte aligned");
  const int64_t nvec = std::max<int64_t>(1, numel / 8);
  const int grid = (int)std::min<int64_t>(2048, (nvec + 255) / 256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char* sp = static_cast<const char*>(src);
  char* dp = static_cast<char*>(dst);
#define SW_CV(ACT)                                                                    \
  if (to_fp8) convert_fp8_e5m2_kernel<ACT, true><<<grid, 256, 0, s>>>(sp, dp, numel); \
  else convert_fp8_e5m2_kernel<ACT, false><<<grid, 256, 0, s>>>(sp, dp, numel);       \
  break
  switch (act) {
    case SW_F32: SW_CV(float);
    case SW_F16: SW_CV(f16_t);
    default: SW_CV(bf16_t);
  }
#undef SW_CV
  SW_CHECK_LAUNCH();
  return SW_OK;
}

}  // extern "C"
