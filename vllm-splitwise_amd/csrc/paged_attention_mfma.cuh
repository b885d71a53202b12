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
  constexpr bool WIDE = FP8;
  // the 8-wave form (small launches, two waves per SIMD at most) keeps the widening of an fp8 piece beside the
  // products that use it; with 4 waves the compiler's own order is the faster one
  constexpr bool PIN = FP8 && NW == 8;
  constexpr int DT = HEAD / 16;            // 16-dim tiles of O^T
  static_assert(HEAD % 64 == 0 && (BS == 16 || BS == 32), "head size / block size of the matrix-core decode kernel");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  S* p_s = reinterpret_cast<S*>(smem);                                            // [NW][16][kPaMfmaPRow]
  float* acc_s = reinterpret_cast<float*>(smem + NW * kPaMfmaHeads * kPaMfmaPRow * 2);   // [NW][qg][HEAD]
  float* m_s = acc_s + NW * qg * HEAD;                                            // [NW][qg]
  float* l_s = m_s + NW * qg;                                                     // [NW][qg]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  // One 64-token group of a wave: its block-table entries, its K / V bytes as loaded, and the products.
  //   16-bit cache: k[t][ks] / v[p * DT + dt] are the operand fragments themselves
  //       key   (tile t, step ks): token g_tok0 + 16 t + r, dims 32 ks + 8 kq ..+7
  //       value (step p, tile dt): dim 16 dt + r, tokens g_tok0 + 32 p + 8 kq ..+7
  //   fp8 cache: k[t][m] is the 16-byte piece with this lane's slots of k steps 2 m and 2 m + 1, v[dt] the piece
  //       with its slots of both P.V steps (tokens g_tok0 + 16 kq ..+15 of dim 16 dt + r); widened at use
  constexpr int KR = WIDE ? KS / 2 : KS;
  constexpr int VR = WIDE ? DT : 2 * DT;
  constexpr int VT = WIDE ? 1 : 2;
  struct Group {
    int tk[4];                               // block of tokens g_tok0 + 16 t ..+15
    pa_u32x4 k[4][KR], v[VR];
  };
  // a group starts on a multiple of 64 tokens, so each run of 16 tokens lies in one block and the four table
  // entries are wave-uniform: scalar loads, and K base addresses in scalar registers (the wave index above is
  // made uniform for the compiler's sake); runs past the context read the last block's entry
  auto entries = [&](Group& f, int g) {
    const int g_tok0 = tok_begin + (g << 6);
#pragma unroll
    for (int t = 0; t < 4; ++t) f.tk[t] = block_table[min((g_tok0 + 16 * t) / BS, num_ctx_blocks - 1)];
  };
  auto fetch = [&](Group& f, int g) {
    const int g_tok0 = tok_begin + (g << 6);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const char* kp = k_cache + (int64_t)f.tk[t] * block_bytes + head_off;
      const int off = (g_tok0 + 16 * t + r) % BS;
#pragma unroll
      for (int m = 0; m < KR; ++m)           // [HEAD/16][BS][16] bytes (fp8) or [HEAD/8][BS][8] 16-bit elements
        f.k[t][m] = *reinterpret_cast<const pa_u32x4*>(kp + ((4 * m + kq) * BS + off) * 16);
    }
#pragma unroll
    for (int p = 0; p < VT; ++p) {
      const int token = WIDE ? g_tok0 + 16 * kq : g_tok0 + 32 * p + 8 * kq;
      const int run = WIDE ? kq : 2 * p + (kq >> 1);                     // (token - g_tok0) / 16
      const int tv = run == 0 ? f.tk[0] : run == 1 ? f.tk[1] : run == 2 ? f.tk[2] : f.tk[3];
      const char* vp = v_cache + (int64_t)tv * block_bytes + head_off + (token % BS) * CS;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        f.v[p * DT + dt] = *reinterpret_cast<const pa_u32x4*>(vp + (16 * dt + r) * BS * CS);
    }
  };
  auto products = [&](const Group& f, int g) {
    const int g_tok0 = tok_begin + (g << 6);
    // ---------------------------------------------------------------- Q.K
    pa_f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = pa_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < KR; ++m) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (WIDE) {
          s[t] = MM::run(e5m2x8_to_f16x8(f.k[t][m][0], f.k[t][m][1]), qf[2 * m], s[t]);
          s[t] = MM::run(e5m2x8_to_f16x8(f.k[t][m][2], f.k[t][m][3]), qf[2 * m + 1], s[t]);
          if (PIN && (t & 1)) __builtin_amdgcn_sched_barrier(0);
        } else {
          s[t] = MM::run(f.k[t][m], qf[m], s[t]);
        }
      }
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
    // tokens past the context may hold anything, NaNs included, and 0 x NaN is NaN: their value elements are
    // cleared (element e of a fragment is real when e < live)
    auto live_mask = [](int live, int w) -> uint32_t {
      return (2 * w < live ? 0x0000ffffu : 0u) | (2 * w + 1 < live ? 0xffff0000u : 0u);
    };
    if constexpr (WIDE) {
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      const pa_u32x4 pf0 = *reinterpret_cast<const pa_u32x4*>(p_w + r * kPaMfmaPRow + 16 * kq);
      const pa_u32x4 pf1 = *reinterpret_cast<const pa_u32x4*>(p_w + r * kPaMfmaPRow + 16 * kq + 8);
      if (g_tok0 + 64 > ctx) {                                               // wave-uniform: the context's last group
        const int live = ctx - (g_tok0 + 16 * kq);
        pa_u32x4 m0, m1;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          m0[w] = live_mask(live, w);
          m1[w] = live_mask(live - 8, w);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          acc[dt] = MM::run(e5m2x8_to_f16x8(f.v[dt][0], f.v[dt][1]) & m0, pf0, acc[dt]);
          acc[dt] = MM::run(e5m2x8_to_f16x8(f.v[dt][2], f.v[dt][3]) & m1, pf1, acc[dt]);
        }
      } else {
#pragma unroll
        for (int d2 = 0; d2 < DT; d2 += 2) {
#pragma unroll
          for (int dt = d2; dt < d2 + 2; ++dt)
            acc[dt] = MM::run(e5m2x8_to_f16x8(f.v[dt][0], f.v[dt][1]), pf0, acc[dt]);
#pragma unroll
          for (int dt = d2; dt < d2 + 2; ++dt)
            acc[dt] = MM::run(e5m2x8_to_f16x8(f.v[dt][2], f.v[dt][3]), pf1, acc[dt]);
          if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if (g_tok0 + 32 * p >= tok_end) break;                               // wave-uniform
        const pa_u32x4 pf = *reinterpret_cast<const pa_u32x4*>(p_w + r * kPaMfmaPRow + 32 * p + 8 * kq);
        if (g_tok0 + 32 * p + 32 > ctx) {
          const int live = ctx - (g_tok0 + 32 * p + 8 * kq);
          pa_u32x4 mk;
#pragma unroll
          for (int w = 0; w < 4; ++w) mk[w] = live_mask(live, w);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) acc[dt] = MM::run(f.v[p * DT + dt] & mk, pf, acc[dt]);
        } else {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) acc[dt] = MM::run(f.v[p * DT + dt], pf, acc[dt]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                  // the strip is rewritten next group
  };

  for (int g = wave; g < n_groups; g += NW) {
    Group a;
    entries(a, g);
    fetch(a, g);
    products(a, g);
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
