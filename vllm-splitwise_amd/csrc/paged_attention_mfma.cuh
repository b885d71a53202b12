// Decode attention for grouped-query layouts on the matrix cores (included by paged_attention.cuh).
//
// Semantics are those of paged_attention_kernel (csrc/attention/attention_kernels.cu:83-417 of the
// reference: fp32 logits, probabilities rounded to the activation dtype before P.V, fp32
// accumulation, 1e-6 in the denominator, V of tokens past the context read as zero).  What differs
// is where the multiply-adds run.  With G query heads sharing a KV head, every K and V byte is
// used G times; on the vector unit (one token per lane, dot2) a workgroup that serves 8 heads issues
// eight dot instructions and two LDS broadcasts of q per 16 loaded bytes and is bound by those, not
// by the loads.  Here a wave forms
//     S [16 tokens x 16 heads]  = K tile [16 tokens x HEAD] . Q^T [HEAD x 16 heads]
//     O^T [16 dims x 16 heads] += V^T tile [16 dims x 32 tokens] . P^T [32 tokens x 16 heads]
// with v_mfma_f32_16x16x32_{bf16,f16}.  Both cache layouts ARE operand layouts: a lane's A fragment
// is 16 contiguous bytes of one token's key ([head/8, block, 8]: 8 dims of token `lane & 15`) or of
// one dim's value row ([head, block]: 8 tokens of dim `lane & 15`), so K and V go from memory to
// the matrix core without a shuffle, an LDS image or an unpack.  The head index is the lane's
// column (lane & 15) in S, in P and in O^T alike, so the running maximum, the rescale and the
// denominator are lane-local; only P changes hands (a lane has 4 tokens x 4 tiles of one head after
// Q.K and needs 8 consecutive tokens for P.V): 2 KiB per 64 tokens through a wave-private LDS strip.
// Up to 16 query heads per workgroup; unused head columns hold zeros and are never written out.
// An fp8 (e5m2) cache is widened to f16 by a byte move (e5m2 is the high byte of a half) and the
// products run as f16 whatever the activation dtype: bf16 queries and probabilities are re-expressed
// in f16 (exact except below 2^-14, where the f16 step of 6e-8 is far below the 1e-2 tolerance of
// the fp8 path).  bf16 reaches 3e38 where f16 ends at 65504: a head whose largest |q| exceeds 2^15 is
// scaled down by a power of two first (exact) and the power goes into that head's softmax scale, so
// large queries give the scores the reference's fp32 products give
// (csrc/quantization/fp8_e5m2_kvcache/quant_utils.cuh:73-81 widens fp8 to bf16 and multiplies in fp32).
#pragma once

namespace sw {

using pa_f32x4 = __attribute__((ext_vector_type(4))) float;
using pa_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using pa_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using pa_u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

template <bool F16> struct PaMfma;
template <> struct PaMfma<true> {
  __device__ static __forceinline__ pa_f32x4 run(pa_u32x4 a, pa_u32x4 b, pa_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pa_f16x8, a), __builtin_bit_cast(pa_f16x8, b), c,
                                                  0, 0, 0);
  }
  __device__ static __forceinline__ uint32_t pack(float lo, float hi) {
    return (uint32_t)float_to_f16_bits(lo) | ((uint32_t)float_to_f16_bits(hi) << 16);
  }
};
template <> struct PaMfma<false> {
  __device__ static __forceinline__ pa_f32x4 run(pa_u32x4 a, pa_u32x4 b, pa_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pa_bf16x8, a), __builtin_bit_cast(pa_bf16x8, b),
                                                   c, 0, 0, 0);
  }
  __device__ static __forceinline__ uint32_t pack(float lo, float hi) {
    return (uint32_t)float_to_bf16_bits(lo) | ((uint32_t)float_to_bf16_bits(hi) << 16);
  }
};

// 8 e5m2 bytes (two dwords) -> 8 halves (four dwords): each byte becomes the high byte of its half
__device__ __forceinline__ pa_u32x4 e5m2x8_to_f16x8(uint32_t lo, uint32_t hi) {
  return pa_u32x4{__builtin_amdgcn_perm(0u, lo, 0x010c000cu), __builtin_amdgcn_perm(0u, lo, 0x030c020cu),
                  __builtin_amdgcn_perm(0u, hi, 0x010c000cu), __builtin_amdgcn_perm(0u, hi, 0x030c020cu)};
}

constexpr int kPaMfmaHeads = 16;        // head columns of one workgroup
constexpr int kPaMfmaPRow = 72;         // 16-bit elements per row of the probability strip (64 + padding)

// NW waves per workgroup share one (sequence, head group[, partition]): 64-token groups go to the waves
// round robin.  A launch of few workgroups takes more waves each, so that every SIMD of the chip
// still has a wave streaming (the launcher picks 4 or 8).
template <typename act_t, bool FP8, int HEAD, int BS, int NW>
__global__ void __launch_bounds__(NW * 64)
paged_attention_mfma_kernel(void* __restrict__ out_, float* __restrict__ exp_sums, float* __restrict__ max_logits,
                            const void* __restrict__ q_, const void* __restrict__ k_cache_,
                            const void* __restrict__ v_cache_, int num_heads, int num_kv_heads, float scale,
                            const int32_t* __restrict__ block_tables, const int32_t* __restrict__ context_lens,
                            int max_num_blocks_per_seq, const float* __restrict__ alibi_slopes, int64_t q_stride,
                            int64_t kv_block_stride, int64_t kv_head_stride, int partitioned, int qg) {
  using S = uint16_t;
  constexpr bool F16M = FP8 || std::is_same<act_t, f16_t>::value;        // element type of the products
  constexpr bool REEXPRESS = FP8 && std::is_same<act_t, bf16_t>::value;  // bf16 values carried as f16
  using MM = PaMfma<F16M>;
  constexpr int CS = FP8 ? 1 : 2;
  constexpr int KS = HEAD / 32;            // k steps of Q.K
  // An fp8 cache holds 16 cache elements in its 16-byte pieces where a 16-bit cache holds 8, so a lane that takes
  // "its 8 elements" of a k step loads 8 bytes: half the bytes in flight per load instruction, which is what bounds
  // small launches.  A matrix product does not care in which order the contraction index is walked as long as both
  // operands agree, so with WIDE the lane takes a whole 16-byte piece -- 16 head elements of a key (two k steps'
  // worth for this lane), 16 tokens of a value row -- and the OTHER operand (q fragments from global memory, P
  // fragments from the strip) is addressed to match:
  //   Q.K   step 2 m + h, slot (kq, j)  <->  head element 64 m + 16 kq + 8 h + j      (not 32 ks + 8 kq + j)
  //   P.V   step p,       slot (kq, j)  <->  token g_tok0 + 16 kq + 8 p + j           (not g_tok0 + 32 p + 8 kq + j)
  constexpr bool WIDE = FP8 && HEAD % 64 == 0;
  constexpr int DT = HEAD / 16;            // 16-dim tiles of O^T
  static_assert(HEAD % 32 == 0 && (BS == 16 || BS == 32), "head size / block size of the matrix-core decode kernel");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  S* p_s = reinterpret_cast<S*>(smem);                                            // [NW][16][kPaMfmaPRow]
  float* acc_s = reinterpret_cast<float*>(smem + NW * kPaMfmaHeads * kPaMfmaPRow * 2);   // [NW][qg][HEAD]
  float* m_s = acc_s + NW * qg * HEAD;                                            // [NW][qg]
  float* l_s = m_s + NW * qg;                                                     // [NW][qg]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 15;                 // K: token of the tile; V: dim of the tile; everywhere else: head column
  const int kq = lane >> 4;                // quarter of the 32-wide k step this lane feeds

  // all head groups of a sequence on one XCD, as in paged_attention_kernel
  int seq = blockIdx.y, hx = blockIdx.x;
  {
    const int total = gridDim.x * gridDim.y;
    if ((total & 7) == 0 && total >= 1024) {
      const int lin = blockIdx.y * gridDim.x + blockIdx.x;
      if ((gridDim.y & 7) == 0) {                            // XCD x: sequences x, x + 8, ... (not a contiguous eighth)
        const int local = lin >> 3;
        const int local_seq = local / (int)gridDim.x;
        hx = local - local_seq * (int)gridDim.x;
        seq = local_seq * 8 + (lin & 7);
      } else {
        const int work = (lin & 7) * (total >> 3) + (lin >> 3);
        seq = work / (int)gridDim.x;
        hx = work - seq * (int)gridDim.x;
      }
    }
  }
  const int part_idx = blockIdx.z;
  const int num_parts = gridDim.z;
  const int groups_per_kv = num_heads / num_kv_heads;
  const int wg_per_kv = groups_per_kv / qg;
  const int kv_head = hx / wg_per_kv;
  const int head0 = kv_head * groups_per_kv + (hx - kv_head * wg_per_kv) * qg;

  const int ctx = context_lens[seq];
  const int tok_begin = partitioned ? part_idx * kPartitionSize : 0;
  if (partitioned && tok_begin >= ctx) return;
  const int tok_end = partitioned ? min(tok_begin + kPartitionSize, ctx) : ctx;
  const int num_ctx_blocks = (ctx + BS - 1) / BS;

  // ---- Q^T fragments: lane (head r, dims 32 ks + 8 kq ..+7); zero for the unused head columns
  pa_u32x4 qf[KS];
  float head_scale = scale;                // softmax scale of the lane's head column
  {
    const S* q = reinterpret_cast<const S*>(q_) + seq * q_stride + (int64_t)(head0 + r) * HEAD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      pa_u32x4 v = {0u, 0u, 0u, 0u};
      const int d0 = WIDE ? 64 * (ks >> 1) + 16 * kq + 8 * (ks & 1) : 32 * ks + 8 * kq;
      if (r < qg) v = *reinterpret_cast<const pa_u32x4*>(q + d0);
      qf[ks] = v;
    }
    if constexpr (REEXPRESS) {
      float amax = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int w = 0; w < 4; ++w)
          amax = fmaxf(amax, fmaxf(fabsf(bf16_bits_to_float((uint16_t)(qf[ks][w] & 0xffffu))),
                                   fabsf(bf16_bits_to_float((uint16_t)(qf[ks][w] >> 16)))));
      amax = fmaxf(amax, __shfl_xor(amax, 16, 64));          // the four lanes of a head column hold its HEAD dims
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      float down = 1.f;
      if (amax > 32768.f && amax < INFINITY) {
        int e;
        (void)frexpf(amax, &e);                               // amax = m * 2^e, 0.5 <= m < 1
        down = ldexpf(1.f, 15 - e);                           // |q| * down <= 2^15
        head_scale = scale * ldexpf(1.f, e - 15);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int w = 0; w < 4; ++w)
          qf[ks][w] = PaMfma<true>::pack(bf16_bits_to_float((uint16_t)(qf[ks][w] & 0xffffu)) * down,
                                         bf16_bits_to_float((uint16_t)(qf[ks][w] >> 16)) * down);
    }
  }
  const float slope = (alibi_slopes && r < qg) ? alibi_slopes[head0 + r] : 0.f;

  const char* k_cache = reinterpret_cast<const char*>(k_cache_);
  const char* v_cache = reinterpret_cast<const char*>(v_cache_);
  const int32_t* block_table = block_tables + (int64_t)seq * max_num_blocks_per_seq;
  const int64_t head_off = (int64_t)kv_head * kv_head_stride * CS;
  const int64_t block_bytes = kv_block_stride * CS;

  float m_run = -FLT_MAX, l_lane = 0.f;
  pa_f32x4 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc[dt] = pa_f32x4{0.f, 0.f, 0.f, 0.f};

  S* p_w = p_s + wave * kPaMfmaHeads * kPaMfmaPRow;
  const int n_groups = (tok_end - tok_begin + 63) >> 6;

  for (int g = wave; g < n_groups; g += NW) {
    const int g_tok0 = tok_begin + (g << 6);

    // ------------------------------------------------ loads: 4 key tiles, 2 x DT value fragments
    // key fragment (tile t, step ks): token g_tok0 + 16 t + r, dims 32 ks + 8 kq ..+7
    pa_u32x4 kf[4][KS];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int token = g_tok0 + 16 * t + r;
      const int bidx = min(token / BS, num_ctx_blocks - 1);               // keep the table read in range
      const char* kp = k_cache + (int64_t)block_table[bidx] * block_bytes + head_off;
      const int off = token % BS;
      if constexpr (WIDE) {                  // [HEAD/16][BS][16] bytes: a whole piece = this lane's slots of two k steps
#pragma unroll
        for (int m = 0; m < KS / 2; ++m) {
          const pa_u32x4 w = *reinterpret_cast<const pa_u32x4*>(kp + ((4 * m + kq) * BS + off) * 16);
          kf[t][2 * m] = e5m2x8_to_f16x8(w[0], w[1]);
          kf[t][2 * m + 1] = e5m2x8_to_f16x8(w[2], w[3]);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if constexpr (FP8) {               // the lane's 8 dims are half a 16-byte piece
            const uint2 w = *reinterpret_cast<const uint2*>(kp + ((2 * ks + (kq >> 1)) * BS + off) * 16 + 8 * (kq & 1));
            kf[t][ks] = e5m2x8_to_f16x8(w.x, w.y);
          } else {                           // [HEAD/8][BS][8] 16-bit elements
            kf[t][ks] = *reinterpret_cast<const pa_u32x4*>(kp + ((4 * ks + kq) * BS + off) * 16);
          }
        }
      }
    }
    // value fragment (step p, dim tile dt): dim 16 dt + r, tokens g_tok0 + 32 p + 8 kq ..+7
    // (WIDE: tokens g_tok0 + 16 kq + 8 p ..+7 -- one 16-byte load brings both steps' tokens)
    pa_u32x4 vf[2][DT];
    if constexpr (WIDE) {
      const int token = g_tok0 + 16 * kq;
      const int bidx = min(token / BS, num_ctx_blocks - 1);
      const char* vp = v_cache + (int64_t)block_table[bidx] * block_bytes + head_off + (token % BS);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const pa_u32x4 w = *reinterpret_cast<const pa_u32x4*>(vp + (16 * dt + r) * BS);
        vf[0][dt] = e5m2x8_to_f16x8(w[0], w[1]);
        vf[1][dt] = e5m2x8_to_f16x8(w[2], w[3]);
      }
      if (g_tok0 + 64 > ctx) {
        // tokens past the context may hold anything, NaNs included, and 0 x NaN is NaN: clear them
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int live = ctx - (token + 8 * p);                             // elements 0 .. live-1 are real
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const uint32_t mask = (2 * w < live ? 0x0000ffffu : 0u) | (2 * w + 1 < live ? 0xffff0000u : 0u);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vf[p][dt][w] &= mask;
          }
        }
      }
    } else {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int token = g_tok0 + 32 * p + 8 * kq;
      const int bidx = min(token / BS, num_ctx_blocks - 1);
      const char* vp = v_cache + (int64_t)block_table[bidx] * block_bytes + head_off + (token % BS) * CS;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        if constexpr (FP8) {
          const uint2 w = *reinterpret_cast<const uint2*>(vp + (16 * dt + r) * BS);
          vf[p][dt] = e5m2x8_to_f16x8(w.x, w.y);
        } else {
          vf[p][dt] = *reinterpret_cast<const pa_u32x4*>(vp + (16 * dt + r) * BS * 2);
        }
      }
      if (g_tok0 + 32 * p + 32 > ctx) {
        // tokens past the context may hold anything, NaNs included, and 0 x NaN is NaN: clear them
        const int live = ctx - token;                                       // elements 0 .. live-1 are real
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t mask = (2 * w < live ? 0x0000ffffu : 0u) | (2 * w + 1 < live ? 0xffff0000u : 0u);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) vf[p][dt][w] &= mask;
        }
      }
    }
    }

    // ---------------------------------------------------------------- Q.K
    pa_f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = pa_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[t] = MM::run(kf[t][ks], qf[ks], s[t]);
    }

    // ------------------------------------------------------ online softmax
    // lane: head r, tokens g_tok0 + 16 t + 4 kq + i
    float gmax = -FLT_MAX;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int token = g_tok0 + 16 * t + 4 * kq + i;
        float x = s[t][i] * head_scale;
        x += (slope != 0.f) ? slope * (float)(token - ctx + 1) : 0.f;
        x = token < tok_end ? x : -FLT_MAX;
        s[t][i] = x;
        gmax = fmaxf(gmax, x);
      }
    }
    gmax = fmaxf(gmax, __shfl_xor(gmax, 16, 64));
    gmax = fmaxf(gmax, __shfl_xor(gmax, 32, 64));
    const float m_new = fmaxf(m_run, gmax);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float p[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int token = g_tok0 + 16 * t + 4 * kq + i;
        p[i] = token < tok_end ? __expf(s[t][i] - m_new) : 0.f;
        psum += p[i];
        if constexpr (REEXPRESS) p[i] = bf16_bits_to_float(float_to_bf16_bits(p[i]));   // the activation dtype's rounding
      }
      uint2 packed;
      packed.x = MM::pack(p[0], p[1]);
      packed.y = MM::pack(p[2], p[3]);
      *reinterpret_cast<uint2*>(p_w + r * kPaMfmaPRow + 16 * t + 4 * kq) = packed;
    }
    l_lane = l_lane * alpha + psum;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] *= alpha;
    // the strip is wave-private: LDS operations of one wave execute in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    // ---------------------------------------------------------------- P.V
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (!WIDE && g_tok0 + 32 * p >= tok_end) break;                       // wave-uniform (WIDE: a step spans the group)
      const pa_u32x4 pf =
          *reinterpret_cast<const pa_u32x4*>(p_w + r * kPaMfmaPRow + (WIDE ? 16 * kq + 8 * p : 32 * p + 8 * kq));
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) acc[dt] = MM::run(vf[p][dt], pf, acc[dt]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                  // the strip is rewritten next group
  }

  // ------------------------------------------------ merge the four waves
  l_lane += __shfl_xor(l_lane, 16, 64);
  l_lane += __shfl_xor(l_lane, 32, 64);
  if (r < qg) {
    if (kq == 0) {
      m_s[wave * qg + r] = m_run;
      l_s[wave * qg + r] = l_lane;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      *reinterpret_cast<pa_f32x4*>(acc_s + (wave * qg + r) * HEAD + 16 * dt + 4 * kq) = acc[dt];
  }
  __syncthreads();

  for (int i = tid; i < qg * HEAD; i += NW * 64) {
    const int h = i / HEAD;
    const int d = i - h * HEAD;
    float m = -FLT_MAX;
#pragma unroll
    for (int w = 0; w < NW; ++w) m = fmaxf(m, m_s[w * qg + h]);
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float f = __expf(m_s[w * qg + h] - m);
      l = fmaf(l_s[w * qg + h], f, l);
      o = fmaf(acc_s[(w * qg + h) * HEAD + d], f, o);
    }
    const float res = o * __fdividef(1.f, l + 1e-6f);
    const int head = head0 + h;
    S* out = reinterpret_cast<S*>(out_);
    if (partitioned) {
      out[(((int64_t)seq * num_heads + head) * num_parts + part_idx) * HEAD + d] = Conv<act_t>::from_float(res);
      if (d == 0) {
        const int64_t pi = ((int64_t)seq * num_heads + head) * num_parts + part_idx;
        max_logits[pi] = m;
        exp_sums[pi] = l;
      }
    } else {
      out[((int64_t)seq * num_heads + head) * HEAD + d] = Conv<act_t>::from_float(res);
    }
  }
}

}  // namespace sw
