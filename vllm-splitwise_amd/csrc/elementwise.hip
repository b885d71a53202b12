// Row-wise / element-wise operators around the attention hot path: RMSNorm (plain and
// fused with the residual add), rotary position embedding, SiLU-and-multiply.
//
// Semantics: csrc/layernorm_kernels.cu:12-66, csrc/pos_encoding_kernels.cu:10-94,
// csrc/activation_kernels.cu:13-31 of the reference.  All are HBM-bound: 16-byte vector
// accesses, one workgroup per token row, the row kept in registers between the reduction
// and the scaling pass (no second trip to memory).
#include "common.cuh"

#include <algorithm>

namespace sw {

constexpr int kEwThreads = 256;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float total = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return total;
}

// 8 elements (f16/bf16) or 4 elements (f32) per 16-byte vector
template <typename act_t> struct Vec16 {
  using S = typename Conv<act_t>::store_t;
  static constexpr int N = 16 / sizeof(S);
  __device__ static __forceinline__ void load(const S* p, float* f) {
    uint4 raw = *reinterpret_cast<const uint4*>(p);
    const S* e = reinterpret_cast<const S*>(&raw);
#pragma unroll
    for (int i = 0; i < N; ++i) f[i] = Conv<act_t>::to_float(e[i]);
  }
  __device__ static __forceinline__ void store(S* p, const float* f) {
    uint4 raw;
    S* e = reinterpret_cast<S*>(&raw);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = Conv<act_t>::from_float(f[i]);
    *reinterpret_cast<uint4*>(p) = raw;
  }
};

// round-trip through the activation dtype (the reference rounds x * rsqrt(var) to scalar_t
// before multiplying by the weight, layernorm_kernels.cu:34,64)
template <typename act_t>
__device__ __forceinline__ float round_act(float v) {
  return Conv<act_t>::to_float(Conv<act_t>::from_float(v));
}

// FUSED: residual += input (stored), then input = norm(residual) * weight.
// else:  out = norm(input) * weight.
template <typename act_t, bool FUSED, int NV>
__global__ void __launch_bounds__(kEwThreads)
rms_norm_kernel(typename Conv<act_t>::store_t* __restrict__ out,
                typename Conv<act_t>::store_t* __restrict__ input,
                typename Conv<act_t>::store_t* __restrict__ residual,
                const typename Conv<act_t>::store_t* __restrict__ weight, float epsilon, int hidden,
                int64_t in_stride) {
  using V = Vec16<act_t>;
  constexpr int N = V::N;
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  float x[NV][N];
  float ss = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int idx = (v * kEwThreads + threadIdx.x) * N;
    if (idx < hidden) {
      V::load(input + row * in_stride + idx, x[v]);
      if constexpr (FUSED) {
        float r[N];
        V::load(residual + row * hidden + idx, r);
        // the sum of squares uses the unrounded fp32 sum, the stored residual (and the
        // scaling pass) its rounding to the activation dtype (layernorm_kernels.cu:50-65)
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const float sum = x[v][i] + r[i];
          ss = fmaf(sum, sum, ss);
          x[v][i] = round_act<act_t>(sum);
        }
        V::store(residual + row * hidden + idx, x[v]);
      } else {
#pragma unroll
        for (int i = 0; i < N; ++i) ss = fmaf(x[v][i], x[v][i], ss);
      }
    }
  }
  ss = block_sum_256(ss, red);
  const float inv = rsqrtf(ss / (float)hidden + epsilon);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int idx = (v * kEwThreads + threadIdx.x) * N;
    if (idx < hidden) {
      float w[N], y[N];
      V::load(weight + idx, w);
#pragma unroll
      for (int i = 0; i < N; ++i) y[i] = round_act<act_t>(x[v][i] * inv) * w[i];
      if constexpr (FUSED) V::store(input + row * in_stride + idx, y);
      else V::store(out + row * hidden + idx, y);
    }
  }
}

// One workgroup per token.  NEOX: pairs (i, i + rot/2); else GPT-J: pairs (2i, 2i+1).
template <typename act_t, bool NEOX>
__global__ void __launch_bounds__(kEwThreads)
rotary_embedding_kernel(const int64_t* __restrict__ positions,
                        typename Conv<act_t>::store_t* __restrict__ query,
                        typename Conv<act_t>::store_t* __restrict__ key,
                        const typename Conv<act_t>::store_t* __restrict__ cos_sin_cache, int rot_dim,
                        int64_t query_stride, int64_t key_stride, int num_heads, int num_kv_heads,
                        int head_size) {
  using S = typename Conv<act_t>::store_t;
  using V = Vec16<act_t>;
  constexpr int N = V::N;
  const int64_t token = blockIdx.x;
  const int64_t pos = positions[token];
  const int embed = rot_dim / 2;
  const S* cos_p = cos_sin_cache + pos * rot_dim;
  const S* sin_p = cos_p + embed;
  if constexpr (NEOX) {
    const int vec_per_head = embed / N;                 // embed is a multiple of N (checked on the host)
    const int total = (num_heads + num_kv_heads) * vec_per_head;
    for (int i = threadIdx.x; i < total; i += kEwThreads) {
      const int h = i / vec_per_head;
      const int off = (i - h * vec_per_head) * N;
      S* base = (h < num_heads) ? query + token * query_stride + (int64_t)h * head_size
                                : key + token * key_stride + (int64_t)(h - num_heads) * head_size;
      float x[N], y[N], c[N], s[N], ox[N], oy[N];
      V::load(base + off, x);
      V::load(base + embed + off, y);
      V::load(cos_p + off, c);
      V::load(sin_p + off, s);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        ox[j] = x[j] * c[j] - y[j] * s[j];
        oy[j] = y[j] * c[j] + x[j] * s[j];
      }
      V::store(base + off, ox);
      V::store(base + embed + off, oy);
    }
  } else {
    const int total = (num_heads + num_kv_heads) * embed;
    for (int i = threadIdx.x; i < total; i += kEwThreads) {
      const int h = i / embed;
      const int r = i - h * embed;
      S* base = (h < num_heads) ? query + token * query_stride + (int64_t)h * head_size
                                : key + token * key_stride + (int64_t)(h - num_heads) * head_size;
      const float x = Conv<act_t>::to_float(base[2 * r]);
      const float y = Conv<act_t>::to_float(base[2 * r + 1]);
      const float c = Conv<act_t>::to_float(cos_p[r]);
      const float s = Conv<act_t>::to_float(sin_p[r]);
      base[2 * r] = Conv<act_t>::from_float(x * c - y * s);
      base[2 * r + 1] = Conv<act_t>::from_float(y * c + x * s);
    }
  }
}

// RoPE and the cache write of one token in one pass (decode steps: one token per sequence, arbitrary
// slots).  One to four workgroups per token: q is rotated in place; every 16-byte vector of k is rotated, stored
// back in place and into the key cache [block, head, head_size/x, block_size, x]; v is scattered into
// the value cache [block, head, head_size, block_size].  The results equal rotary_embedding followed by
// reshape_and_cache bit for bit: the same fp32 products rounded once to the activation dtype, and an fp8
// cache takes the e5m2 rounding of that rounded value.  A negative slot (padding row) is rotated, not cached.
template <typename act_t, bool FP8, bool NEOX>
__global__ void __launch_bounds__(kEwThreads)
rope_cache_kernel(const int64_t* __restrict__ positions, typename Conv<act_t>::store_t* __restrict__ query,
                  typename Conv<act_t>::store_t* __restrict__ key,
                  const typename Conv<act_t>::store_t* __restrict__ value, char* __restrict__ key_cache,
                  char* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping,
                  const typename Conv<act_t>::store_t* __restrict__ cos_sin_cache, int rot_dim, int64_t query_stride,
                  int64_t key_stride, int64_t value_stride, int num_heads, int num_kv_heads, int head_size,
                  int block_size) {
  using S = typename Conv<act_t>::store_t;
  using V = Vec16<act_t>;
  constexpr int N = V::N;
  constexpr int ES = sizeof(S);
  constexpr int CS = FP8 ? 1 : ES;
  constexpr int X = 16 / CS;
  const int64_t token = blockIdx.x;
  const int first = blockIdx.y * kEwThreads + threadIdx.x;          // gridDim.y workgroups share a token
  const int step = gridDim.y * kEwThreads;
  const int64_t pos = positions[token];
  const int64_t slot = slot_mapping[token];
  const int embed = rot_dim / 2;
  const S* cos_p = cos_sin_cache + pos * rot_dim;
  const S* sin_p = cos_p + embed;
  const int64_t blk = slot >= 0 ? slot / block_size : 0;
  const int boff = slot >= 0 ? (int)(slot - blk * block_size) : 0;
  const int64_t head_bytes = (int64_t)head_size * block_size * CS;
  char* kc = key_cache + blk * head_bytes * num_kv_heads;
  char* vc = value_cache + blk * head_bytes * num_kv_heads;

  // N elements of kv head `h` from head offset `off` on (a multiple of N, N divides x) -> key cache
  auto cache_k = [&](int h, int off, const float* f) {
    if (slot < 0) return;
    char* dst = kc + h * head_bytes + ((int64_t)((off / X) * block_size + boff) * X + (off % X)) * CS;
    if constexpr (!FP8) {
      V::store(reinterpret_cast<S*>(dst), f);
    } else {
      uint32_t w[N / 4];
#pragma unroll
      for (int q = 0; q < N / 4; ++q) {
        uint32_t acc = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc |= (uint32_t)Conv<act_t>::to_e5m2(Conv<act_t>::from_float(f[4 * q + b])) << (8 * b);
        w[q] = acc;
      }
      if constexpr (N == 8) *reinterpret_cast<uint2*>(dst) = make_uint2(w[0], w[1]);
      else *reinterpret_cast<uint32_t*>(dst) = w[0];
    }
  };

  // v first: its stores are the slow part of the launch (2-byte pieces block_size elements apart; the
  // layout allows nothing wider for a single token) and drain while q and k are rotated below.  A lane
  // takes one element per step and neighbouring lanes neighbouring elements, so that the lanes of a store
  // instruction fall into as few 128-byte lines as the layout permits.
  if (slot >= 0) {
    const S* v_row = value + token * value_stride;
    for (int i = first; i < num_kv_heads * head_size; i += step) {
      const int h = i / head_size;
      const int d = i - h * head_size;
      const S e = v_row[i];
      char* dst = vc + h * head_bytes + ((int64_t)d * block_size + boff) * CS;
      if constexpr (FP8) *reinterpret_cast<uint8_t*>(dst) = Conv<act_t>::to_e5m2(e);
      else *reinterpret_cast<S*>(dst) = e;
    }
  }

  const int pass_vecs = (head_size - rot_dim) / N;                  // vectors past the rotated part
  const int per_head = (NEOX ? embed / N : rot_dim / N) + pass_vecs;
  const int rot_items = per_head - pass_vecs;
  const int total = (num_heads + num_kv_heads) * per_head;
  for (int i = first; i < total; i += step) {
    const int h = i / per_head;
    const int r = i - h * per_head;
    const bool is_k = h >= num_heads;
    const int kvh = h - num_heads;
    S* base = is_k ? key + token * key_stride + (int64_t)kvh * head_size : query + token * query_stride + (int64_t)h * head_size;
    if (r >= rot_items) {                                           // not rotated: k still goes to the cache
      if (is_k) {
        const int off = rot_dim + (r - rot_items) * N;
        float x[N];
        V::load(base + off, x);
        cache_k(kvh, off, x);
      }
      continue;
    }
    const int off = r * N;
    if constexpr (NEOX) {
      float x[N], y[N], c[N], sn[N], ox[N], oy[N];
      V::load(base + off, x);
      V::load(base + embed + off, y);
      V::load(cos_p + off, c);
      V::load(sin_p + off, sn);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        ox[j] = x[j] * c[j] - y[j] * sn[j];
        oy[j] = y[j] * c[j] + x[j] * sn[j];
      }
      V::store(base + off, ox);
      V::store(base + embed + off, oy);
      if (is_k) {
        cache_k(kvh, off, ox);
        cache_k(kvh, embed + off, oy);
      }
    } else {
      float x[N], o[N];
      V::load(base + off, x);
#pragma unroll
      for (int j = 0; j < N / 2; ++j) {
        const float c = Conv<act_t>::to_float(cos_p[off / 2 + j]);
        const float sn = Conv<act_t>::to_float(sin_p[off / 2 + j]);
        o[2 * j] = x[2 * j] * c - x[2 * j + 1] * sn;
        o[2 * j + 1] = x[2 * j + 1] * c + x[2 * j] * sn;
      }
      V::store(base + off, o);
      if (is_k) cache_k(kvh, off, o);
    }
  }

}

// out[t, i] = silu(in[t, i]) * in[t, d + i]; silu is rounded to the activation dtype first
// (activation_kernels.cu:22,27-31).
template <typename act_t>
__global__ void __launch_bounds__(kEwThreads)
silu_and_mul_kernel(typename Conv<act_t>::store_t* __restrict__ out,
                    const typename Conv<act_t>::store_t* __restrict__ input, int64_t num_tokens, int d,
                    int64_t in_stride) {
  using V = Vec16<act_t>;
  constexpr int N = V::N;
  const int vec_per_row = d / N;
  const int64_t total = num_tokens * vec_per_row;
  for (int64_t i = (int64_t)blockIdx.x * kEwThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kEwThreads) {
    const int64_t t = i / vec_per_row;
    const int off = (int)(i - t * vec_per_row) * N;
    float x[N], y[N], o[N];
    V::load(input + t * in_stride + off, x);
    V::load(input + t * in_stride + d + off, y);
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = round_act<act_t>(x[j] / (1.0f + expf(-x[j]))) * y[j];
    V::store(out + t * d + off, o);
  }
}

template <typename act_t, bool FUSED>
int launch_rms(void* out, void* input, void* residual, const void* weight, float eps, int64_t num_tokens,
               int hidden, int64_t in_stride, hipStream_t stream) {
  using S = typename Conv<act_t>::store_t;
  constexpr int N = 16 / sizeof(S);
  const int per_pass = kEwThreads * N;
  const int nv = (hidden + per_pass - 1) / per_pass;
  S* o = static_cast<S*>(out);
  S* in = static_cast<S*>(input);
  S* res = static_cast<S*>(residual);
  const S* w = static_cast<const S*>(weight);
  dim3 grid((unsigned)num_tokens);
  switch (nv) {
    case 1: rms_norm_kernel<act_t, FUSED, 1><<<grid, kEwThreads, 0, stream>>>(o, in, res, w, eps, hidden, in_stride); break;
    case 2: rms_norm_kernel<act_t, FUSED, 2><<<grid, kEwThreads, 0, stream>>>(o, in, res, w, eps, hidden, in_stride); break;
    case 3: case 4: rms_norm_kernel<act_t, FUSED, 4><<<grid, kEwThreads, 0, stream>>>(o, in, res, w, eps, hidden, in_stride); break;
    case 5: case 6: case 7: case 8:
      rms_norm_kernel<act_t, FUSED, 8><<<grid, kEwThreads, 0, stream>>>(o, in, res, w, eps, hidden, in_stride); break;
    case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16:
      rms_norm_kernel<act_t, FUSED, 16><<<grid, kEwThreads, 0, stream>>>(o, in, res, w, eps, hidden, in_stride); break;
    default:
      set_error("rms_norm: hidden size %d too large", hidden);
      return SW_ERR_INVALID;
  }
  SW_CHECK_LAUNCH();
  return SW_OK;
}

static bool act_ok(int dt) { return dt == SW_F32 || dt == SW_F16 || dt == SW_BF16; }

}  // namespace sw

extern "C" {

int sw_rms_norm(void* out, const void* input, const void* weight, float epsilon, int64_t num_tokens,
                int hidden_size, int64_t input_stride, int act_dtype, void* stream) {
  using namespace sw;
  if (num_tokens == 0) return SW_OK;
  SW_CHECK_ARG(act_ok(act_dtype), "rms_norm: unsupported dtype %d", act_dtype);
  const int es = dtype_size(act_dtype);
  SW_CHECK_ARG(hidden_size > 0 && (hidden_size * es) % 16 == 0 && (input_stride * es) % 16 == 0,
               "rms_norm: hidden size %d must be a multiple of 16 bytes", hidden_size);
  hipStream_t s = static_cast<hipStream_t>(stream);
  void* in = const_cast<void*>(input);
  switch (act_dtype) {
    case SW_F32: return launch_rms<float, false>(out, in, nullptr, weight, epsilon, num_tokens, hidden_size, input_stride, s);
    case SW_F16: return launch_rms<f16_t, false>(out, in, nullptr, weight, epsilon, num_tokens, hidden_size, input_stride, s);
    default: return launch_rms<bf16_t, false>(out, in, nullptr, weight, epsilon, num_tokens, hidden_size, input_stride, s);
  }
}

int sw_fused_add_rms_norm(void* input, void* residual, const void* weight, float epsilon,
                          int64_t num_tokens, int hidden_size, int64_t input_stride, int act_dtype,
                          void* stream) {
  using namespace sw;
  if (num_tokens == 0) return SW_OK;
  SW_CHECK_ARG(act_ok(act_dtype), "fused_add_rms_norm: unsupported dtype %d", act_dtype);
  const int es = dtype_size(act_dtype);
  SW_CHECK_ARG(hidden_size > 0 && (hidden_size * es) % 16 == 0 && (input_stride * es) % 16 == 0,
               "fused_add_rms_norm: hidden size %d must be a multiple of 16 bytes", hidden_size);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (act_dtype) {
    case SW_F32: return launch_rms<float, true>(nullptr, input, residual, weight, epsilon, num_tokens, hidden_size, input_stride, s);
    case SW_F16: return launch_rms<f16_t, true>(nullptr, input, residual, weight, epsilon, num_tokens, hidden_size, input_stride, s);
    default: return launch_rms<bf16_t, true>(nullptr, input, residual, weight, epsilon, num_tokens, hidden_size, input_stride, s);
  }
}

int sw_rotary_embedding(const int64_t* positions, void* query, void* key, int64_t num_tokens,
                        int num_heads, int num_kv_heads, int head_size, const void* cos_sin_cache,
                        int rot_dim, int is_neox, int64_t query_stride, int64_t key_stride,
                        int act_dtype, void* stream) {
  using namespace sw;
  if (num_tokens == 0) return SW_OK;
  SW_CHECK_ARG(act_ok(act_dtype), "rotary_embedding: unsupported dtype %d", act_dtype);
  const int es = dtype_size(act_dtype);
  SW_CHECK_ARG(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "rotary_embedding: bad rot_dim %d", rot_dim);
  if (is_neox) {
    SW_CHECK_ARG(((rot_dim / 2) * es) % 16 == 0 && (head_size * es) % 16 == 0 &&
                     (query_stride * es) % 16 == 0 && (key_stride * es) % 16 == 0,
                 "rotary_embedding: rot_dim/2 = %d must span a multiple of 16 bytes", rot_dim / 2);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((unsigned)num_tokens);
#define SW_ROPE(ACT)                                                                              \
  do {                                                                                            \
    using S = Conv<ACT>::store_t;                                                                 \
    if (is_neox)                                                                                  \
      rotary_embedding_kernel<ACT, true><<<grid, kEwThreads, 0, s>>>(                             \
          positions, static_cast<S*>(query), static_cast<S*>(key), static_cast<const S*>(cos_sin_cache), \
          rot_dim, query_stride, key_stride, num_heads, num_kv_heads, head_size);                 \
    else                                                                                          \
      rotary_embedding_kernel<ACT, false><<<grid, kEwThreads, 0, s>>>(                            \
          positions, static_cast<S*>(query), static_cast<S*>(key), static_cast<const S*>(cos_sin_cache), \
          rot_dim, query_stride, key_stride, num_heads, num_kv_heads, head_size);                 \
  } while (0)
  switch (act_dtype) {
    case SW_F32: SW_ROPE(float); break;
    case SW_F16: SW_ROPE(f16_t); break;
    default: SW_ROPE(bf16_t); break;
  }
#undef SW_ROPE
  SW_CHECK_LAUNCH();
  return SW_OK;
}

int sw_rotary_embedding_and_cache(const int64_t* positions, void* query, void* key, const void* value,
                                  void* key_cache, void* value_cache, const int64_t* slot_mapping,
                                  int64_t num_tokens, int num_heads, int num_kv_heads, int head_size,
                                  const void* cos_sin_cache, int rot_dim, int is_neox, int block_size, int x,
                                  int64_t query_stride, int64_t key_stride, int64_t value_stride, int act_dtype,
                                  int cache_dtype, void* stream) {
  using namespace sw;
  if (num_tokens == 0) return SW_OK;
  SW_CHECK_ARG(act_ok(act_dtype), "rotary_embedding_and_cache: unsupported dtype %d", act_dtype);
  const bool fp8 = cache_dtype == SW_FP8_E5M2;
  SW_CHECK_ARG(fp8 || cache_dtype == act_dtype, "Unsupported data type of kv cache: %d", cache_dtype);
  const int es = dtype_size(act_dtype);
  const int n = 16 / es;
  SW_CHECK_ARG(num_heads > 0 && num_kv_heads > 0 && head_size > 0 && block_size > 0, "rotary_embedding_and_cache: bad shape");
  SW_CHECK_ARG(x == 16 / (fp8 ? 1 : es), "rotary_embedding_and_cache: x=%d but 16/sizeof(cache element)=%d", x,
               16 / (fp8 ? 1 : es));
  SW_CHECK_ARG(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "rotary_embedding_and_cache: bad rot_dim %d", rot_dim);
  SW_CHECK_ARG(head_size % n == 0 && (head_size - rot_dim) % n == 0 && (is_neox ? (rot_dim / 2) % n == 0 : rot_dim % n == 0),
               "rotary_embedding_and_cache: head_size %d / rot_dim %d do not divide into 16-byte vectors", head_size,
               rot_dim);
  SW_CHECK_ARG((query_stride * es) % 16 == 0 && (key_stride * es) % 16 == 0 && (value_stride * es) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(query) % 16 == 0 && reinterpret_cast<uintptr_t>(key) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(value) % 16 == 0 && reinterpret_cast<uintptr_t>(cos_sin_cache) % 16 == 0 &&
                   (!is_neox || ((rot_dim / 2) * es) % 16 == 0),
               "rotary_embedding_and_cache: rows of query / key / value must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // a token's elements are spread over up to four workgroups while that keeps every thread busy
  const int64_t per_token = (int64_t)num_kv_heads * head_size;
  int parts = (int)std::min<int64_t>(4, std::max<int64_t>(1, per_token / kEwThreads));
  dim3 grid((unsigned)num_tokens, (unsigned)parts);
#define SW_RC_LAUNCH(ACT, F8, NX)                                                                              \
  rope_cache_kernel<ACT, F8, NX><<<grid, kEwThreads, 0, s>>>(                                                  \
      positions, static_cast<Conv<ACT>::store_t*>(query), static_cast<Conv<ACT>::store_t*>(key),               \
      static_cast<const Conv<ACT>::store_t*>(value), static_cast<char*>(key_cache), static_cast<char*>(value_cache), \
      slot_mapping, static_cast<const Conv<ACT>::store_t*>(cos_sin_cache), rot_dim, query_stride, key_stride,  \
      value_stride, num_heads, num_kv_heads, head_size, block_size)
#define SW_RC_ACT(ACT)                                                                                         \
  do {                                                                                                         \
    if (fp8) { if (is_neox) SW_RC_LAUNCH(ACT, true, true); else SW_RC_LAUNCH(ACT, true, false); }              \
    else { if (is_neox) SW_RC_LAUNCH(ACT, false, true); else SW_RC_LAUNCH(ACT, false, false); }                \
  } while (0)
  switch (act_dtype) {
    case SW_F32: SW_RC_ACT(float); break;
    case SW_F16: SW_RC_ACT(f16_t); break;
    default: SW_RC_ACT(bf16_t); break;
  }
#undef SW_RC_ACT
#undef SW_RC_LAUNCH
  SW_CHECK_LAUNCH();
  return SW_OK;
}

int sw_silu_and_mul(void* out, const void* input, int64_t num_tokens, int d, int64_t input_stride,
                    int act_dtype, void* stream) {
  using namespace sw;
  if (num_tokens == 0) return SW_OK;
  SW_CHECK_ARG(act_ok(act_dtype), "silu_and_mul: unsupported dtype %d", act_dtype);
  const int es = dtype_size(act_dtype);
  SW_CHECK_ARG(d > 0 && (d * es) % 16 == 0 && (input_stride * es) % 16 == 0,
               "silu_and_mul: d = %d must span a multiple of 16 bytes", d);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t total = num_tokens * (d / (16 / es));
  const int grid = (int)std::min<int64_t>(4096, (total + kEwThreads - 1) / kEwThreads);
#define SW_SILU(ACT)                                                                                   \
  silu_and_mul_kernel<ACT><<<grid, kEwThreads, 0, s>>>(static_cast<Conv<ACT>::store_t*>(out),          \
                                                       static_cast<const Conv<ACT>::store_t*>(input),  \
                                                       num_tokens, d, input_stride)
  switch (act_dtype) {
    case SW_F32: SW_SILU(float); break;
    case SW_F16: SW_SILU(f16_t); break;
    default: SW_SILU(bf16_t); break;
  }
#undef SW_SILU
  SW_CHECK_LAUNCH();
  return SW_OK;
}

}  // extern "C"
