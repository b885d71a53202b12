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
