// Single-query (decode) attention over a paged KV cache, for gfx950 wave64.
//
// Semantics: csrc/attention/attention_kernels.cu:83-417 (v1 / v2 partition
// kernel) and :484-575 (v2 reduce) of the reference: fp32 logits, masked
// tokens excluded, probabilities rounded to the activation dtype before P.V,
// fp32 accumulation, 1e-6 added to the softmax denominator, out-of-context V
// lanes zeroed.  The tiling is new:
//
//   * one workgroup (4 waves) per (sequence, kv head [x group of QG query
//     heads], partition).  All QG query heads that share a kv head are served
//     from ONE pass over K and V, so GQA caches are read once, not QG times.
//   * a wave walks the context in groups of 64 tokens, ONE TOKEN PER LANE for
//     Q.K: lane t streams its token's key with 16-byte loads (for the
//     [head/x, block, x] layout 16 lanes cover 256 contiguous bytes, and the
//     head/x loads of a lane together consume whole 4-KiB [block, head] tiles).
//     The dot product is lane-local: no cross-lane reduction per token.
//   * softmax is online per wave (running max, one wave-max per 64 tokens;
//     the denominator stays a per-lane partial until the end).
//   * P.V re-maps lanes onto 16-byte pieces of the [head, block] V tile in
//     memory order (fully contiguous 1-KiB wave loads); the 64 probabilities
//     travel through a wave-private LDS strip in the activation dtype.
//   * the four waves merge (max, sum, acc) through LDS once at the end.
// Nothing here is GEMM-shaped enough to leave the HBM roofline: no MFMA.
#pragma once

#include "common.cuh"
#include <float.h>
#include <stdlib.h>

namespace sw {

constexpr int kPaThreads = 256;
constexpr int kPaWaves = 4;
constexpr int kPaLeanUpTo = 4096;     // workgroups: launches up to this size take the vector kernel's LEAN form
constexpr int kPartitionSize = 512;   // vllm/.../ops/paged_attn.py:12 _PARTITION_SIZE

template <typename act_t, bool FP8>
struct CacheTraits {
  using S = typename Conv<act_t>::store_t;
  static constexpr int ES = sizeof(S);
  static constexpr int CS = FP8 ? 1 : ES;
  static constexpr int X = 16 / CS;
};

// largest divisor of n that is <= cap
constexpr int key_batch(int n, int cap = 16) {
  int b = n < cap ? n : cap;
  while (n % b) --b;
  return b;
}

// N cache elements packed in dwords w -> fp32
template <typename act_t, bool FP8, int N>
__device__ __forceinline__ void unpack_cache(const uint32_t* w, float* out) {
  using S = typename Conv<act_t>::store_t;
  if constexpr (FP8) {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = e5m2_to_float((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
  } else if constexpr (sizeof(S) == 2) {
#pragma unroll
    for (int i = 0; i < N; ++i)
      out[i] = Conv<act_t>::to_float(static_cast<uint16_t>(w[i >> 1] >> (16 * (i & 1))));
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __uint_as_float(w[i]);
  }
}

template <typename act_t, int N>
__device__ __forceinline__ void unpack_act(const uint32_t* w, float* out) {
  unpack_cache<act_t, false, N>(w, out);
}

// Packed 16-bit path: q, k, p and v stay as (even, odd) element pairs in one dword and go through
// v_dot2_f32_f16 / v_dot2c_f32_bf16 (two multiply-adds per instruction, fp32 accumulation), so
// no element is ever unpacked to fp32.
using h16x2 = __attribute__((ext_vector_type(2))) _Float16;
using b16x2 = __attribute__((ext_vector_type(2))) __bf16;

template <typename act_t> struct Dot2 { static constexpr bool kAvailable = false; };
template <> struct Dot2<f16_t> {
  static constexpr bool kAvailable = true;
  __device__ static __forceinline__ float run(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, a), __builtin_bit_cast(h16x2, b), c, false);
  }
  __device__ static __forceinline__ void settle(float&) {}
};
template <> struct Dot2<bf16_t> {
  static constexpr bool kAvailable = true;
  __device__ static __forceinline__ float run(uint32_t a, uint32_t b, float c) {
    // The instruction itself, not __builtin_amdgcn_fdot2_f32_bf16: with the builtin the scheduler
    // hoists so many loads ahead of the dot products that the 4- and 8-head kernels need more
    // than 300 registers (one wave per SIMD, spills), where the f16 twin needs 130-160.
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
    return c;
  }
  // A dot result may be read by a different VALU or memory instruction only 3 wait states after
  // the dot issued, and overwritten only after 4; the hardware does not interlock this and the
  // compiler, which inserts the s_nop for its own dot instructions, cannot see into the asm above.
  // Call this on the last result of a chain before anything but another dot2c touches it.
  __device__ static __forceinline__ void settle(float& c) { asm volatile("s_nop 3" : "+v"(c)); }
};

// N cache elements (N/2 pairs) -> dwords of activation-dtype pairs.  fp8 e5m2 is the high byte of
// a half, so a byte moves to the high half of its 16-bit lane (f16 activations only).
template <bool FP8, int N>
__device__ __forceinline__ void cache_pairs(const uint32_t* w, uint32_t* out) {
  if constexpr (FP8) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      out[2 * i] = __builtin_amdgcn_perm(0u, w[i], 0x010c000cu);       // bytes [0, b0, 0, b1]
      out[2 * i + 1] = __builtin_amdgcn_perm(0u, w[i], 0x030c020cu);   // bytes [0, b2, 0, b3]
    }
  } else {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) out[i] = w[i];
  }
}

// 16-bit cache of the activation dtype, or an fp8 cache under f16 activations (bf16 has no exact
// byte move from e5m2 and keeps the fp32 path)
template <typename act_t, bool FP8> struct PackedPath { static constexpr bool value = Dot2<act_t>::kAvailable && !FP8; };
template <> struct PackedPath<f16_t, true> { static constexpr bool value = true; };

// LEAN: the group's block-table entries are read once (the P.V phase takes block b's entry from lane b * BS of
// the Q.K phase's read instead of reading the table again in front of every block's loads) and the wave index
// is handed to the compiler as uniform, which moves the P.V address arithmetic to the scalar unit.  Faster
// wherever a wave's chain of dependent loads shows -- launches of a few thousand workgroups or fewer (8 x 1024
// tokens 24.5 -> 22 us, 4 x 1024 20.8 -> 16.6, 128 x 860 311 -> 306) and every fp8 launch (256 x 860: 325 ->
// 305 us) -- and 0.4-1.6 % slower on the largest 16-bit launches, which keep the plain form.
template <typename act_t, bool FP8, int HEAD, int BS, int QG, bool LEAN>
__global__ void __launch_bounds__(kPaThreads)
paged_attention_kernel(void* __restrict__ out_,            // v1: [S,H,HEAD]; v2: tmp_out [S,H,P,HEAD]
                       float* __restrict__ exp_sums,       // v2 only [S,H,P]
                       float* __restrict__ max_logits,     // v2 only [S,H,P]
                       const void* __restrict__ q_, const void* __restrict__ k_cache_,
                       const void* __restrict__ v_cache_, int num_heads, int num_kv_heads,
                       float scale, const int32_t* __restrict__ block_tables,
                       const int32_t* __restrict__ context_lens, int max_num_blocks_per_seq,
                       const float* __restrict__ alibi_slopes, int64_t q_stride,
                       int64_t kv_block_stride, int64_t kv_head_stride, int partitioned) {
  using T = CacheTraits<act_t, FP8>;
  using S = typename T::S;
  constexpr int ES = T::ES, CS = T::CS, X = T::X;
  constexpr int NCH = HEAD / X;                    // 16-byte key pieces per token
  // key pieces in flight per batch; several query heads per workgroup leave fewer registers
  constexpr int KB = key_batch(NCH, QG >= 4 ? 4 : 16);
  constexpr int NB = 64 / BS;                      // cache blocks per 64-token group
  constexpr int VT = (16 / CS) < BS ? (16 / CS) : BS;   // tokens per V piece
  constexpr int VBYTES = VT * CS;                  // 16 (or 8: fp8 cache, block 8)
  constexpr int UPR = BS / VT;                     // V pieces per cache row
  constexpr int UNITS = HEAD * UPR;                // V pieces per [block, head] tile
  constexpr int NIT = (UNITS + 63) / 64;           // V pieces per lane per block
  constexpr int PW = VT * ES / 4;                  // dwords of probabilities per V piece
  constexpr bool PACKED = PackedPath<act_t, FP8>::value;
  static_assert(HEAD % X == 0 && NCH % KB == 0, "head size must split into key batches");
  static_assert(64 % BS == 0 && BS % VT == 0, "block size must divide the 64-token group");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* q_s = reinterpret_cast<float*>(smem);                              // [QG][HEAD]
  S* p_s = reinterpret_cast<S*>(smem + QG * HEAD * 4);                      // [4][QG][64]
  float* red_s = reinterpret_cast<float*>(smem + QG * HEAD * 4 + kPaWaves * QG * 64 * ES);
  // red_s: [4][QG][HEAD] acc, then [4][QG] m, then [4][QG] l

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = LEAN ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
  // Workgroups go to the 8 XCDs round robin in launch order.  All head groups of a sequence walk
  // the same block table and the same cache blocks (a block holds its heads side by side), so
  // they are renumbered to run on ONE XCD, next to each other in time: its L2 and its address
  // translation then see each block of the sequence once, not once per XCD (2-3 % on launches
  // of thousands of workgroups; small launches keep the plain order).  XCD x takes the sequences x, x + 8,
  // x + 16, ... when their number is a multiple of 8 -- not a contiguous eighth of the batch: batches are
  // often ordered by age, hence by length, and a contiguous eighth would hand one XCD all the long contexts
  // (a length-sorted long-tailed batch ran at half the bandwidth that way).
  int seq = blockIdx.y, hx = blockIdx.x;
  {
    const int total = gridDim.x * gridDim.y;
    if ((total & 7) == 0 && total >= 1024) {
      const int lin = blockIdx.y * gridDim.x + blockIdx.x;
      if ((gridDim.y & 7) == 0) {
        const int local = lin >> 3;                          // this XCD's local-th workgroup
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
  const int groups_per_kv = num_heads / num_kv_heads;       // query heads per kv head
  const int wg_per_kv = groups_per_kv / QG;
  const int kv_head = hx / wg_per_kv;
  const int head0 = kv_head * groups_per_kv + (hx - kv_head * wg_per_kv) * QG;

  const int ctx = context_lens[seq];
  const int tok_begin = partitioned ? part_idx * kPartitionSize : 0;
  if (partitioned && tok_begin >= ctx) return;              // uniform: whole workgroup leaves
  const int tok_end = partitioned ? min(tok_begin + kPartitionSize, ctx) : ctx;
  const int num_ctx_blocks = (ctx + BS - 1) / BS;

  // ---- query -> LDS as fp32, pre-multiplied by nothing (scale is applied to the dot)
  const uint32_t* q_p = reinterpret_cast<const uint32_t*>(smem);           // packed path: [QG][HEAD/2] pairs
  {
    const S* q = reinterpret_cast<const S*>(q_) + seq * q_stride + (int64_t)head0 * HEAD;
    if constexpr (PACKED) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(smem);
      const uint32_t* src = reinterpret_cast<const uint32_t*>(q);          // 4-byte aligned: checked on the host
      for (int i = tid; i < QG * HEAD / 2; i += kPaThreads) dst[i] = src[i];
    } else {
      for (int i = tid; i < QG * HEAD; i += kPaThreads) q_s[i] = Conv<act_t>::to_float(q[i]);
    }
  }
  __syncthreads();

  const char* k_cache = reinterpret_cast<const char*>(k_cache_);
  const char* v_cache = reinterpret_cast<const char*>(v_cache_);
  const int32_t* block_table = block_tables + (int64_t)seq * max_num_blocks_per_seq;

  float slope[QG];
#pragma unroll
  for (int h = 0; h < QG; ++h) slope[h] = alibi_slopes ? alibi_slopes[head0 + h] : 0.f;

  float m_run[QG], l_lane[QG], acc[QG][NIT];
#pragma unroll
  for (int h = 0; h < QG; ++h) {
    m_run[h] = -FLT_MAX;
    l_lane[h] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) acc[h][it] = 0.f;
  }

  S* p_w = p_s + wave * QG * 64;
  const int n_groups = (tok_end - tok_begin + 63) >> 6;
  const int blk_in_group = lane / BS;
  const int off_in_blk = lane % BS;

  for (int g = wave; g < n_groups; g += kPaWaves) {
    const int g_tok0 = tok_begin + (g << 6);
    const int token = g_tok0 + lane;
    const bool valid = token < tok_end;
    // lane l holds the table entry of its token's block (blocks past the context repeat the last entry)
    const int my_entry = block_table[min(g_tok0 / BS + blk_in_group, num_ctx_blocks - 1)];

    // ---------------------------------------------------------------- Q.K
    float qk[QG];
#pragma unroll
    for (int h = 0; h < QG; ++h) qk[h] = 0.f;
    {
      const int64_t phys = my_entry;
      const char* kp = k_cache + (phys * kv_block_stride + (int64_t)kv_head * kv_head_stride) * CS +
                       off_in_blk * 16;
#pragma unroll
      for (int cb = 0; cb < NCH; cb += KB) {
        uint4 kraw[KB];
#pragma unroll
        for (int c = 0; c < KB; ++c)
          kraw[c] = *reinterpret_cast<const uint4*>(kp + (int64_t)(cb + c) * BS * 16);
#pragma unroll
        for (int c = 0; c < KB; ++c) {
          if constexpr (PACKED) {
            uint32_t kp2[X / 2];
            cache_pairs<FP8, X>(reinterpret_cast<const uint32_t*>(&kraw[c]), kp2);
#pragma unroll
            for (int h = 0; h < QG; ++h) {
              const uint32_t* qh = q_p + h * (HEAD / 2) + (cb + c) * (X / 2);
#pragma unroll
              for (int e = 0; e < X / 2; e += 4) {
                const uint4 qv = *reinterpret_cast<const uint4*>(qh + e);     // LDS broadcast
                qk[h] = Dot2<act_t>::run(qv.x, kp2[e], qk[h]);
                qk[h] = Dot2<act_t>::run(qv.y, kp2[e + 1], qk[h]);
                qk[h] = Dot2<act_t>::run(qv.z, kp2[e + 2], qk[h]);
                qk[h] = Dot2<act_t>::run(qv.w, kp2[e + 3], qk[h]);
              }
            }
          } else {
          float kf[X];
          unpack_cache<act_t, FP8, X>(reinterpret_cast<const uint32_t*>(&kraw[c]), kf);
#pragma unroll
          for (int h = 0; h < QG; ++h) {
            const float* qh = q_s + h * HEAD + (cb + c) * X;
#pragma unroll
            for (int e = 0; e < X; e += 4) {
              const float4 qv = *reinterpret_cast<const float4*>(qh + e);   // LDS broadcast
              qk[h] = fmaf(qv.x, kf[e], qk[h]);
              qk[h] = fmaf(qv.y, kf[e + 1], qk[h]);
              qk[h] = fmaf(qv.z, kf[e + 2], qk[h]);
              qk[h] = fmaf(qv.w, kf[e + 3], qk[h]);
            }
          }
          }
        }
      }
    }

    // ------------------------------------------------------ online softmax
#pragma unroll
    for (int h = 0; h < QG; ++h) {
      if constexpr (PACKED) Dot2<act_t>::settle(qk[h]);
      float s = qk[h] * scale;
      s += (slope[h] != 0.f) ? slope[h] * (float)(token - ctx + 1) : 0.f;
      const float gmax = wave_max(valid ? s : -FLT_MAX);
      const float m_new = fmaxf(m_run[h], gmax);
      const float alpha = __expf(m_run[h] - m_new);
      m_run[h] = m_new;
      float p = valid ? __expf(s - m_new) : 0.f;
      l_lane[h] = l_lane[h] * alpha + p;
#pragma unroll
      for (int it = 0; it < NIT; ++it) acc[h][it] *= alpha;
      p_w[h * 64 + lane] = Conv<act_t>::from_float(p);      // rounded to the activation dtype
    }
    // p_w is wave-private: LDS ops of one wave execute in order, no barrier needed.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    // ---------------------------------------------------------------- P.V
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int btok0 = g_tok0 + b * BS;
      if (btok0 >= tok_end) break;                          // wave-uniform
      const int bidx = btok0 / BS;
      const int64_t phys = LEAN ? __builtin_amdgcn_readlane(my_entry, b * BS) : block_table[bidx];
      const char* vp = v_cache + (phys * kv_block_stride + (int64_t)kv_head * kv_head_stride) * CS;
      const bool last_block = (bidx == num_ctx_blocks - 1);
      uint32_t vraw[NIT][4];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int u = it * 64 + lane;
        if (UNITS % 64 == 0 || u < UNITS) {
          if constexpr (VBYTES == 16) {
            const uint4 t = *reinterpret_cast<const uint4*>(vp + (int64_t)u * 16);
            vraw[it][0] = t.x; vraw[it][1] = t.y; vraw[it][2] = t.z; vraw[it][3] = t.w;
          } else {
            const uint2 t = *reinterpret_cast<const uint2*>(vp + (int64_t)u * 8);
            vraw[it][0] = t.x; vraw[it][1] = t.y; vraw[it][2] = 0; vraw[it][3] = 0;
          }
        } else {
          vraw[it][0] = vraw[it][1] = vraw[it][2] = vraw[it][3] = 0;
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int u = it * 64 + lane;
        const int part = u % UPR;                           // which VT-token slice of the row
        if constexpr (PACKED) {
          uint32_t vp2[VT / 2];
          cache_pairs<FP8, VT>(vraw[it], vp2);
          if (last_block) {
            // zero V of tokens past the context (they may hold NaNs); their p is 0 too
#pragma unroll
            for (int j = 0; j < VT / 2; ++j) {
              const int tok = btok0 + part * VT + 2 * j;
              vp2[j] &= (tok < ctx ? 0x0000ffffu : 0u) | (tok + 1 < ctx ? 0xffff0000u : 0u);
            }
          }
#pragma unroll
          for (int h = 0; h < QG; ++h) {
            const uint32_t* pp = reinterpret_cast<const uint32_t*>(p_w + h * 64 + b * BS + part * VT);
            float a = acc[h][it];
            if constexpr (VT / 2 >= 4) {
#pragma unroll
              for (int w4 = 0; w4 < VT / 2; w4 += 4) {
                const uint4 t = *reinterpret_cast<const uint4*>(pp + w4);
                a = Dot2<act_t>::run(t.x, vp2[w4], a);
                a = Dot2<act_t>::run(t.y, vp2[w4 + 1], a);
                a = Dot2<act_t>::run(t.z, vp2[w4 + 2], a);
                a = Dot2<act_t>::run(t.w, vp2[w4 + 3], a);
              }
            } else {
#pragma unroll
              for (int w = 0; w < VT / 2; ++w) a = Dot2<act_t>::run(pp[w], vp2[w], a);
            }
            acc[h][it] = a;
          }
        } else {
        float vf[VT];
        unpack_cache<act_t, FP8, VT>(vraw[it], vf);
        if (last_block) {
          // zero V of tokens past the context (they may hold NaNs); their p is 0 too
#pragma unroll
          for (int j = 0; j < VT; ++j) vf[j] = (btok0 + part * VT + j < ctx) ? vf[j] : 0.f;
        }
#pragma unroll
        for (int h = 0; h < QG; ++h) {
          const uint32_t* pp =
              reinterpret_cast<const uint32_t*>(p_w + h * 64 + b * BS + part * VT);
          uint32_t praw[PW];
#pragma unroll
          for (int w4 = 0; w4 < PW; w4 += 4) {
            if constexpr (PW >= 4) {
              const uint4 t = *reinterpret_cast<const uint4*>(pp + w4);
              praw[w4] = t.x; praw[w4 + 1] = t.y; praw[w4 + 2] = t.z; praw[w4 + 3] = t.w;
            }
          }
          if constexpr (PW == 2) {
            const uint2 t = *reinterpret_cast<const uint2*>(pp);
            praw[0] = t.x; praw[1] = t.y;
          }
          float pf[VT];
          unpack_act<act_t, VT>(praw, pf);
          float a = acc[h][it];
#pragma unroll
          for (int j = 0; j < VT; ++j) a = fmaf(pf[j], vf[j], a);
          acc[h][it] = a;
        }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // p_w is rewritten next group
  }

  // ------------------------------------------------ merge the four waves
  // (inside the loop the next reader of acc, the rescale by alpha, sits behind the whole Q.K phase)
  if constexpr (PACKED) {
#pragma unroll
    for (int h = 0; h < QG; ++h)
#pragma unroll
      for (int it = 0; it < NIT; ++it) Dot2<act_t>::settle(acc[h][it]);
  }
  float* acc_s = red_s;                                     // [4][QG][HEAD]
  float* m_s = red_s + kPaWaves * QG * HEAD;                // [4][QG]
  float* l_s = m_s + kPaWaves * QG;                         // [4][QG]
#pragma unroll
  for (int h = 0; h < QG; ++h) {
    const float l = wave_sum(l_lane[h]);
    if (lane == 0) {
      m_s[wave * QG + h] = m_run[h];
      l_s[wave * QG + h] = l;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      float a = acc[h][it];
#pragma unroll
      for (int sft = 1; sft < UPR; sft <<= 1) a += __shfl_xor(a, sft, 64);
      const int u = it * 64 + lane;
      if ((u % UPR) == 0 && u < UNITS) acc_s[(wave * QG + h) * HEAD + u / UPR] = a;
    }
  }
  __syncthreads();

  for (int i = tid; i < QG * HEAD; i += kPaThreads) {
    const int h = i / HEAD;
    const int d = i - h * HEAD;
    float m = -FLT_MAX;
#pragma unroll
    for (int w = 0; w < kPaWaves; ++w) m = fmaxf(m, m_s[w * QG + h]);
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < kPaWaves; ++w) {
      const float f = __expf(m_s[w * QG + h] - m);
      l = fmaf(l_s[w * QG + h], f, l);
      o = fmaf(acc_s[(w * QG + h) * HEAD + d], f, o);
    }
    const float r = o * __fdividef(1.f, l + 1e-6f);
    const int head = head0 + h;
    S* out = reinterpret_cast<S*>(out_);
    if (partitioned) {
      out[(((int64_t)seq * num_heads + head) * num_parts + part_idx) * HEAD + d] =
          Conv<act_t>::from_float(r);
      if (d == 0) {
        const int64_t pi = ((int64_t)seq * num_heads + head) * num_parts + part_idx;
        max_logits[pi] = m;
        exp_sums[pi] = l;
      }
    } else {
      out[((int64_t)seq * num_heads + head) * HEAD + d] = Conv<act_t>::from_float(r);
    }
  }
}

// Grid (num_heads, num_seqs); combines the partitions of one (seq, head).
template <typename act_t, int HEAD>
__global__ void __launch_bounds__(128)
paged_attention_reduce_kernel(void* __restrict__ out_, const float* __restrict__ exp_sums,
                              const float* __restrict__ max_logits,
                              const void* __restrict__ tmp_out_,
                              const int32_t* __restrict__ context_lens, int max_num_partitions) {
  using S = typename Conv<act_t>::store_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* w_s = reinterpret_cast<float*>(smem);               // [num_partitions] weights
  __shared__ float red[4];
  const int head = blockIdx.x, seq = blockIdx.y, num_heads = gridDim.x;
  const int ctx = context_lens[seq];
  const int np = (ctx + kPartitionSize - 1) / kPartitionSize;
  const int64_t base = ((int64_t)seq * num_heads + head) * max_num_partitions;
  const S* tmp = reinterpret_cast<const S*>(tmp_out_) + base * HEAD;
  S* out = reinterpret_cast<S*>(out_) + ((int64_t)seq * num_heads + head) * HEAD;
  const int tid = threadIdx.x;
  if (np <= 1) {
    if (np == 1)
      for (int i = tid; i < HEAD; i += 128) out[i] = tmp[i];
    else
      for (int i = tid; i < HEAD; i += 128) out[i] = Conv<act_t>::from_float(0.f);
    return;
  }
  float m = -FLT_MAX;
  for (int i = tid; i < np; i += 128) m = fmaxf(m, max_logits[base + i]);
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(red[0], red[1]);
  float s = 0.f;
  for (int i = tid; i < np; i += 128) {
    const float r = exp_sums[base + i] * expf(max_logits[base + i] - m);
    w_s[i] = r;
    s += r;
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[2 + (tid >> 6)] = s;
  __syncthreads();
  const float inv = __fdividef(1.f, red[2] + red[3] + 1e-6f);
  for (int d = tid; d < HEAD; d += 128) {
    float a = 0.f;
    for (int j = 0; j < np; ++j) a += Conv<act_t>::to_float(tmp[(int64_t)j * HEAD + d]) * w_s[j] * inv;
    out[d] = Conv<act_t>::from_float(a);
  }
}

struct PaArgs {
  void* out; float* exp_sums; float* max_logits; void* tmp_out;
  const void* q; const void* k; const void* v;
  int num_seqs, num_heads, num_kv_heads, head_size; float scale;
  const int32_t* block_tables; const int32_t* context_lens;
  int block_size, max_context_len, max_num_blocks_per_seq;
  const float* alibi; int64_t q_stride, kv_block_stride, kv_head_stride;
  int partitioned; hipStream_t stream;
};

template <typename act_t, bool FP8, int HEAD, int BS, int QG>
int launch_pa(const PaArgs& a) {
  using T = CacheTraits<act_t, FP8>;
  const int g = a.num_heads / a.num_kv_heads;
  const int parts = a.partitioned ? (a.max_context_len + kPartitionSize - 1) / kPartitionSize : 1;
  dim3 grid(a.num_kv_heads * (g / QG), a.num_seqs, parts);
  const size_t smem = (size_t)QG * HEAD * 4 + (size_t)kPaWaves * QG * 64 * T::ES +
                      (size_t)kPaWaves * QG * HEAD * 4 + 2 * kPaWaves * QG * 4;
  // fp8 caches always take the lean form (there is no other instantiation for them)
  const bool lean = FP8 || (int64_t)grid.x * grid.y * grid.z <= kPaLeanUpTo;
  auto kernel = lean ? paged_attention_kernel<act_t, FP8, HEAD, BS, QG, true>
                     : paged_attention_kernel<act_t, FP8, HEAD, BS, QG, FP8>;
  kernel<<<grid, kPaThreads, smem, a.stream>>>(
      a.partitioned ? a.tmp_out : a.out, a.exp_sums, a.max_logits, a.q, a.k, a.v, a.num_heads,
      a.num_kv_heads, a.scale, a.block_tables, a.context_lens, a.max_num_blocks_per_seq, a.alibi,
      a.q_stride, a.kv_block_stride, a.kv_head_stride, a.partitioned);
  SW_CHECK_LAUNCH();
  if (a.partitioned) {
    dim3 rgrid(a.num_heads, a.num_seqs);
    paged_attention_reduce_kernel<act_t, HEAD><<<rgrid, 128, parts * sizeof(float), a.stream>>>(
        a.out, a.exp_sums, a.max_logits, a.tmp_out, a.context_lens, parts);
    SW_CHECK_LAUNCH();
  }
  return SW_OK;
}

}  // namespace sw
#include "paged_attention_mfma.cuh"
namespace sw {

template <typename act_t, bool FP8, int HEAD, int BS, int NW>
int launch_pa_mfma_nw(const PaArgs& a, int qg, dim3 grid) {
  const size_t smem = (size_t)NW * kPaMfmaHeads * kPaMfmaPRow * 2 + (size_t)NW * qg * HEAD * 4 + 2 * NW * qg * 4;
  auto kernel = paged_attention_mfma_kernel<act_t, FP8, HEAD, BS, NW>;
  if (smem > 64 * 1024) {
    // on every launch: the attribute belongs to the function ON THE CURRENT DEVICE, and a process may drive
    // several devices from several threads; the call costs well under a microsecond
    SW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  }
  kernel<<<grid, NW * 64, smem, a.stream>>>(
      a.partitioned ? a.tmp_out : a.out, a.exp_sums, a.max_logits, a.q, a.k, a.v, a.num_heads, a.num_kv_heads,
      a.scale, a.block_tables, a.context_lens, a.max_num_blocks_per_seq, a.alibi, a.q_stride, a.kv_block_stride,
      a.kv_head_stride, a.partitioned, qg);
  SW_CHECK_LAUNCH();
  return SW_OK;
}

template <typename act_t, bool FP8, int HEAD, int BS>
int launch_pa_mfma(const PaArgs& a, int qg) {
  const int g = a.num_heads / a.num_kv_heads;
  const int parts = a.partitioned ? (a.max_context_len + kPartitionSize - 1) / kPartitionSize : 1;
  dim3 grid(a.num_kv_heads * (g / qg), a.num_seqs, parts);
  // every SIMD of the 256 compute units wants a wave: a launch of few workgroups takes 8 waves each
  // (16 would leave a wave 128 registers, which this kernel's fragments do not fit)
  const int64_t wgs = (int64_t)grid.x * grid.y * grid.z;
  const int rc = wgs >= 256 ? launch_pa_mfma_nw<act_t, FP8, HEAD, BS, 4>(a, qg, grid)
                            : launch_pa_mfma_nw<act_t, FP8, HEAD, BS, 8>(a, qg, grid);
  if (rc != SW_OK) return rc;
  if (a.partitioned) {
    dim3 rgrid(a.num_heads, a.num_seqs);
    paged_attention_reduce_kernel<act_t, HEAD><<<rgrid, 128, parts * sizeof(float), a.stream>>>(
        a.out, a.exp_sums, a.max_logits, a.tmp_out, a.context_lens, parts);
    SW_CHECK_LAUNCH();
  }
  return SW_OK;
}

// Query heads per workgroup of the matrix-core kernel: the largest divisor of the group size that
// fits its 16 head columns; below 4 the vector kernels are the better fit (0 = do not use it).
inline int pa_mfma_width(int group, int head_size, int block_size, int act_bytes, const void* q, int64_t q_stride,
                         int min_width = 4) {
  if (act_bytes != 2 || (head_size != 64 && head_size != 128) || (block_size != 16 && block_size != 32)) return 0;
  if (reinterpret_cast<uintptr_t>(q) % 16 != 0 || q_stride % 8 != 0) return 0;     // 16-byte query fragments
  int qg = group < kPaMfmaHeads ? group : kPaMfmaHeads;
  while (group % qg) --qg;
  return qg >= min_width ? qg : 0;
}

// Dispatch over head size / block size / query-group width for one (act, cache) pair.
// Query-group widths above 1 are built for head sizes 64 and 128 and blocks of 16 / 32, up to
// MAXQG query heads per workgroup: the widest width whose kernel still fits the register file
// without spilling (f16: 8; bf16: 4, its two-address dot instruction costs registers; bf16 over an
// fp8 cache, which takes the fp32 path: 2).
template <typename act_t, bool FP8, int MAXQG>
int dispatch_pa(const PaArgs& a) {
  const int g = a.num_heads / a.num_kv_heads;
  if constexpr (!std::is_same<act_t, float>::value) {
    // bf16 activations over an fp8 cache have no packed vector path (e5m2 widens to f16, not bf16, by a byte
    // move): the matrix-core kernel, which re-expresses q and p in f16, takes them whatever the group size
    constexpr bool kNoPackedPath = FP8 && std::is_same<act_t, bf16_t>::value;
    const int qgm = pa_mfma_width(g, a.head_size, a.block_size, 2, a.q, a.q_stride, kNoPackedPath ? 1 : 4);
    if (qgm) {
      if (a.head_size == 64) {
        if (a.block_size == 16) return launch_pa_mfma<act_t, FP8, 64, 16>(a, qgm);
        return launch_pa_mfma<act_t, FP8, 64, 32>(a, qgm);
      }
      if (a.block_size == 16) return launch_pa_mfma<act_t, FP8, 128, 16>(a, qgm);
      return launch_pa_mfma<act_t, FP8, 128, 32>(a, qgm);
    }
  }
  int qg = 1;
  if (MAXQG > 1 && (a.head_size == 64 || a.head_size == 128) && a.block_size >= 16) {
    if (g % 8 == 0) qg = 8;
    else if (g % 4 == 0) qg = 4;
    else if (g % 2 == 0) qg = 2;
    while (qg > MAXQG) qg /= 2;
  }
#define SW_PA_BS(HEAD, QG)                                                       \
  switch (a.block_size) {                                                        \
    case 8: if (QG == 1) return launch_pa<act_t, FP8, HEAD, 8, 1>(a); break;     \
    case 16: return launch_pa<act_t, FP8, HEAD, 16, QG>(a);                      \
    case 32: return launch_pa<act_t, FP8, HEAD, 32, QG>(a);                      \
    default: break;                                                              \
  }                                                                              \
  set_error("Unsupported block size: %d", a.block_size);                         \
  return SW_ERR_INVALID;
#define SW_PA_QG(HEAD)                                        \
  if constexpr (MAXQG >= 8) { if (qg == 8) { SW_PA_BS(HEAD, 8) } } \
  if constexpr (MAXQG >= 4) { if (qg == 4) { SW_PA_BS(HEAD, 4) } } \
  if constexpr (MAXQG >= 2) { if (qg == 2) { SW_PA_BS(HEAD, 2) } } \
  { SW_PA_BS(HEAD, 1) }
  switch (a.head_size) {
    case 64: { SW_PA_QG(64) }
    case 80: { SW_PA_BS(80, 1) }
    case 96: { SW_PA_BS(96, 1) }
    case 112: { SW_PA_BS(112, 1) }
    case 128: { SW_PA_QG(128) }
    case 256: { SW_PA_BS(256, 1) }
    default: break;
  }
#undef SW_PA_QG
#undef SW_PA_BS
  set_error("Unsupported head size: %d", a.head_size);
  return SW_ERR_INVALID;
}

// one translation unit per (activation, cache) pair provides these
int pa_dispatch_f16(const PaArgs& a);
int pa_dispatch_f16_fp8(const PaArgs& a);
int pa_dispatch_bf16(const PaArgs& a);
int pa_dispatch_bf16_fp8(const PaArgs& a);
int pa_dispatch_f32(const PaArgs& a);
int pa_dispatch_f32_fp8(const PaArgs& a);

}  // namespace sw
