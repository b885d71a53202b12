// Weight-only int4 linear layers (GPTQ and AWQ checkpoints) with fp16 activations on gfx950.
//
// What it replaces
//   GPTQ: csrc/quantization/gptq/q_gemm.cu -- gptq_shuffle (:2748-2769, shuffle_exllama_weight
//         :2032-2093, make_sequential_4bit_kernel :1812-1853, shuffle_4bit_8 qdq_4.cuh:16-35),
//         gptq_gemm (:2712-2746 -> gemm_half_q_half_cuda :1687-1754), the reconstruct kernels
//         (:920-1037 shuffled weights, :1517-1561 plain weights + g_idx).
//   AWQ:  csrc/quantization/awq/gemm_kernels.cu -- awq_gemm (:648-705), awq_dequantize
//         (:585-640, kernel :274-335), nibble unpack dequantize.cuh:17-84.
//
// Value semantics (both formats): W[k][n] = fp16( (q[k][n] - zero[g][n]) * scale[g][n] ) -- the
// integer difference is exact, the product is rounded once to fp16 exactly as the reference's
// reconstruct / dequantize kernels do -- and C = A . W with fp32 accumulation on the matrix
// cores, rounded once to fp16.  GPTQ stores zero - 1 (q_gemm.cu:277), AWQ stores zero.
//
// The fused GEMMs are built for the weight-streaming regime (decode, a few to a few hundred
// rows of A): every int4 word is read from HBM exactly once per 32*MT rows of A, unpacked in
// registers with the 0x6400 exponent trick and fed straight into v_mfma_f32_32x32x16_f16 as
// the B operand:
//   * GPTQ packs 8 consecutive k of one column in a word -- exactly one lane's B fragment.  A
//     lane loads 16 bytes = 4 adjacent columns, a wave covers 128 columns x 16 k per step, and
//     column n of MFMA c is n0 + 4*lane + c, so the lane's results are 4 adjacent columns too.
//   * AWQ packs 8 columns of one k in a word; a wave writes the unpacked 16 k x 256 column
//     fp16 tile to its private LDS image and gets B fragments back through the transposing
//     ds_read_b64_tr_b16.
//   * the 4 waves of a workgroup split the workgroup's k range and add up in LDS; k is split
//     across workgroups as well, each writing an fp32 partial tile, and the LAST workgroup to
//     arrive at a tile (counted with one atomic) adds the partials in a fixed order: one
//     launch, no atomics on data, bitwise reproducible results.
#include "common.cuh"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace sw {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using h2 = __attribute__((ext_vector_type(2))) _Float16;
using i16x4 = __attribute__((ext_vector_type(4))) short;
using lds_i16x4_ptr = __attribute__((address_space(3))) i16x4*;

constexpr int kCounterInts = 4096;                  // head of the workspace: one arrival counter per tile
constexpr int kCounterBytes = kCounterInts * 4;
constexpr uint32_t kMagic = 0x64006400u;            // half2(1024, 1024)

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
constexpr int kAuxSystem = 17;                      // sc0 | sc1: through every cache level, both ways

// Partial tiles cross workgroups (and therefore XCDs, whose L2s do not snoop each other).  A
// release/acquire fence pair would write back and invalidate a whole L2 per workgroup; instead
// the partials themselves are stored and loaded through the caches, the stores are waited for,
// and the arrival counter is a relaxed device-scope atomic.
__device__ __forceinline__ int arrive(int32_t* counter) {
  return __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ h2 as_h2(uint32_t u) { return __builtin_bit_cast(h2, u); }
__device__ __forceinline__ uint32_t as_u32(h2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ h2 splat(_Float16 v) { return h2{v, v}; }

// ------------------------------------------------------------------ GPTQ nibble handling
// A shuffled word holds rows q0 q2 q4 q6 in its low half and q1 q3 q5 q7 in its high half
// (qdq_4.cuh:12-14), so masking with 0x000f000f pairs rows (0,1), with 0x00f000f0 rows (2,3)
// scaled by 16, and the same again after a shift by 8.
struct GptqCol {
  h2 z1;     // -(1024 + zero)
  h2 z16;    // -(64 + zero)
  h2 s;      // scale, both halves
};

__device__ __forceinline__ GptqCol gptq_col(uint32_t zero, uint16_t scale_bits) {
  GptqCol c;
  const uint32_t zb = 0xe400u | zero;               // -(1024 + zero) for zero <= 16
  c.z1 = as_h2(zb | (zb << 16));
  c.z16 = splat((_Float16)(-64 - (int)zero));
  c.s = as_h2((uint32_t)scale_bits | ((uint32_t)scale_bits << 16));
  return c;
}

// 8 rows of one column -> fp16 weights in row order (4 dwords = one MFMA B fragment)
__device__ __forceinline__ f16x8 gptq_dequant8(uint32_t w, const GptqCol& col) {
  const h2 y16 = splat((_Float16)0.0625f);
  const h2 q0 = as_h2((w & 0x000f000fu) | kMagic);
  const h2 q1 = as_h2((w & 0x00f000f0u) | kMagic);
  w >>= 8;
  const h2 q2 = as_h2((w & 0x000f000fu) | kMagic);
  const h2 q3 = as_h2((w & 0x00f000f0u) | kMagic);
  const h2 d0 = (q0 + col.z1) * col.s;
  const h2 d1 = (q1 * y16 + col.z16) * col.s;
  const h2 d2 = (q2 + col.z1) * col.s;
  const h2 d3 = (q3 * y16 + col.z16) * col.s;
  union { uint32_t u[4]; f16x8 f; } r;
  r.u[0] = as_u32(d0); r.u[1] = as_u32(d1); r.u[2] = as_u32(d2); r.u[3] = as_u32(d3);
  return r.f;
}

// The same with the masks and the exponent pattern held in registers (lo, hi: scalar; magic: vector), which lets
// each (w & mask) | magic be ONE v_and_or_b32: with literal constants the compiler emits an AND and an OR.
struct DequantRegs { uint32_t lo, hi, magic; };
__device__ __forceinline__ DequantRegs dequant_regs() {
  DequantRegs k{0x000f000fu, 0x00f000f0u, kMagic};
  asm volatile("" : "+s"(k.lo), "+s"(k.hi));
  asm volatile("" : "+v"(k.magic));
  return k;
}
__device__ __forceinline__ f16x8 gptq_dequant8(uint32_t w, const GptqCol& col, const DequantRegs& k) {
  const h2 y16 = splat((_Float16)0.0625f);
  const h2 q0 = as_h2((w & k.lo) | k.magic);
  const h2 q1 = as_h2((w & k.hi) | k.magic);
  w >>= 8;
  const h2 q2 = as_h2((w & k.lo) | k.magic);
  const h2 q3 = as_h2((w & k.hi) | k.magic);
  const h2 d0 = (q0 + col.z1) * col.s;
  const h2 d1 = (q1 * y16 + col.z16) * col.s;
  const h2 d2 = (q2 + col.z1) * col.s;
  const h2 d3 = (q3 * y16 + col.z16) * col.s;
  union { uint32_t u[4]; f16x8 f; } r;
  r.u[0] = as_u32(d0); r.u[1] = as_u32(d1); r.u[2] = as_u32(d2); r.u[3] = as_u32(d3);
  return r.f;
}

__device__ __forceinline__ uint32_t shuffle_nibbles(uint32_t qa) {
  uint32_t qb = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t even = qa & 0x0fu, odd = (qa & 0xf0u) >> 4;
    qa >>= 8;
    qb |= odd << (4 * i + 16);
    qb |= even << (4 * i);
  }
  return qb;
}

// ------------------------------------------------------------------ GPTQ fused GEMM
// A wave streams a run of 16-k steps of one tile of 32*CP columns, keeping PF steps of loads in
// flight in a register ring.
//   lane (j = lane & 31, half = lane >> 5): columns n0 + CP*j .. +CP-1, k rows 8*half..+7 of a step
//   acc[t][c][r] <-> row m0 + 32t + 8(r>>2) + 4*half + (r&3), column n0 + CP*j + c
constexpr int kGemmWaves = 8;

template <int CP> struct WordVec;
template <> struct WordVec<2> { using type = uint2; using scale_t = uint32_t; };
template <> struct WordVec<4> { using type = uint4; using scale_t = uint2; };

template <int MT, int CP, int PF>
__device__ __forceinline__ void gptq_stream(f32x16 (&acc)[MT][CP], const uint16_t* __restrict__ a,
                                            const uint32_t* __restrict__ qweight, const uint32_t* __restrict__ qzeros,
                                            const uint16_t* __restrict__ scales, int M, int N, int K, int group_size,
                                            int m0, int ncl, int p_begin, int p_end, int zoff) {
  // The unit of streaming is a PAIR of 16-k steps (32 k).  Within a pair the k order is permuted
  // so that a lane's two A fragments are 32 contiguous bytes and the two halves of the wave read
  // the whole 64 bytes of a row: lane (j, half), sub-step i covers k = 32 d + 16 half + 8 i .. + 7,
  // i.e. weight word row 4 d + 2 half + i.  A and W use the same order, so the sum is unchanged.
  using wvec = typename WordVec<CP>::type;
  using svec = typename WordVec<CP>::scale_t;
  const int lane = threadIdx.x & 63;
  const int j = lane & 31, half = lane >> 5;
  const int ppg = group_size / 32;                    // pairs per group
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int cc = 0; cc < CP; ++cc)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][cc][r] = 0.f;

  const uint16_t* ap[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) ap[t] = a + (size_t)min(m0 + 32 * t + j, M - 1) * K + 16 * half;
  const uint32_t* wp = qweight + (size_t)(2 * half) * N + ncl;
  const int zshift = 4 * (ncl & 7);                   // this lane's zero nibbles inside the zeros word

  // register ring; slot indices are compile-time so that nothing is spilled
  wvec wq[PF][2];
  f16x8 af[PF][MT][2];
  uint32_t zq[PF];
  svec sq[PF];
  GptqCol col[CP];
  int i_group = p_begin / ppg;                        // group at the issue cursor
  int i_boundary = (i_group + 1) * ppg;               // first pair of the next group, issue side
  int c_boundary = i_boundary;                        // same, consume side

  auto issue = [&](auto SLOT, int d) {
    constexpr int slot = decltype(SLOT)::value;
    if (d >= p_end) return;
    wq[slot][0] = *reinterpret_cast<const wvec*>(wp + (size_t)(4 * d) * N);
    wq[slot][1] = *reinterpret_cast<const wvec*>(wp + (size_t)(4 * d + 1) * N);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      af[slot][t][0] = *reinterpret_cast<const f16x8*>(ap[t] + 32 * d);
      af[slot][t][1] = *reinterpret_cast<const f16x8*>(ap[t] + 32 * d + 8);
    }
    if (d == p_begin || d == i_boundary) {            // wave-uniform: the first pair of a group carries its data
      if (d == i_boundary) { ++i_group; i_boundary += ppg; }
      zq[slot] = qzeros[(size_t)i_group * (N / 8) + ncl / 8];
      sq[slot] = *reinterpret_cast<const svec*>(scales + (size_t)i_group * N + ncl);
    }
  };
  auto consume = [&](auto SLOT, int d) {
    constexpr int slot = decltype(SLOT)::value;
    if (d == p_begin || d == c_boundary) {
      if (d == c_boundary) c_boundary += ppg;
      const uint32_t z = zq[slot] >> zshift;
      if constexpr (CP == 4) {
        col[0] = gptq_col((z & 0xfu) + zoff, (uint16_t)(sq[slot].x & 0xffffu));
        col[1] = gptq_col(((z >> 4) & 0xfu) + zoff, (uint16_t)(sq[slot].x >> 16));
        col[2] = gptq_col(((z >> 8) & 0xfu) + zoff, (uint16_t)(sq[slot].y & 0xffffu));
        col[3] = gptq_col(((z >> 12) & 0xfu) + zoff, (uint16_t)(sq[slot].y >> 16));
      } else {
        col[0] = gptq_col((z & 0xfu) + zoff, (uint16_t)(sq[slot] & 0xffffu));
        col[1] = gptq_col(((z >> 4) & 0xfu) + zoff, (uint16_t)(sq[slot] >> 16));
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint32_t w[CP];
      if constexpr (CP == 4) { w[0] = wq[slot][i].x; w[1] = wq[slot][i].y; w[2] = wq[slot][i].z; w[3] = wq[slot][i].w; }
      else { w[0] = wq[slot][i].x; w[1] = wq[slot][i].y; }
#pragma unroll
      for (int cc = 0; cc < CP; ++cc) {
        const f16x8 b = gptq_dequant8(w[cc], col[cc]);
#pragma unroll
        for (int t = 0; t < MT; ++t)
          acc[t][cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[slot][t][i], b, acc[t][cc], 0, 0, 0);
      }
    }
  };
  static_for<PF>([&](auto SLOT) { issue(SLOT, p_begin + decltype(SLOT)::value); });
  for (int d = p_begin; d < p_end; d += PF) {
    static_for<PF>([&](auto SLOT) {
      constexpr int i = decltype(SLOT)::value;
      if (d + i < p_end) {
        consume(SLOT, d + i);
        issue(SLOT, d + i + PF);
      }
    });
  }
}

// Partial-tile exchange between the workgroups that split k: ROWS accumulator rows per thread.
// `row_of(i)` gives the matrix row of the i-th value set, `val_of(i, c)` its CP column values.
template <int CP, int ROWS, typename ValOf, typename RowOf>
__device__ __forceinline__ void combine_and_store(ValOf val_of, RowOf row_of, bool col_ok, int ncol,
                                                  uint16_t* __restrict__ c, int M, int N, int ksplit,
                                                  float* __restrict__ partials, int32_t* __restrict__ counters,
                                                  int tile, int split, int* s_last) {
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(partials, 0, (int)((size_t)ksplit * M * N * 4), 0x00020000);
  bool last = true;
  if (ksplit > 1) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const int m = row_of(i);
      if (m >= M || !col_ok) continue;
      const int off = (int)((((size_t)split * M + m) * N + ncol) * 4);
      if constexpr (CP == 4) {
        const u32x4 v = {__float_as_uint(val_of(i, 0)), __float_as_uint(val_of(i, 1)), __float_as_uint(val_of(i, 2)),
                         __float_as_uint(val_of(i, 3))};
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, kAuxSystem);
      } else {
        const u32x2 v = {__float_as_uint(val_of(i, 0)), __float_as_uint(val_of(i, 1))};
        __builtin_amdgcn_raw_buffer_store_b64(v, rsrc, off, 0, kAuxSystem);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *s_last = (arrive(&counters[tile]) == ksplit - 1);
    __syncthreads();
    last = *s_last != 0;
  }
  if (!last) return;
  // the final sum, RB rows at a time (all of a thread's loads of one split and block are issued before the first is used)
  constexpr int RB = ROWS > 16 ? 16 : ROWS;              // three RB x CP register sets are live at once
  static_assert(ROWS % RB == 0, "row blocks must tile the thread's rows");
#pragma unroll
  for (int b0 = 0; b0 < ROWS; b0 += RB) {
    float v[RB][CP];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int cc = 0; cc < CP; ++cc) v[i][cc] = ksplit > 1 ? 0.f : val_of(b0 + i, cc);
    if (ksplit > 1) {
      // The splits are added in the fixed order 0, 1, ... whoever arrived last (reproducible); this workgroup's own
      // partial comes from its registers, not from memory, and the loads of split z+1 are in flight while split z
      // is added.
      auto take = [&](int z, float (&p)[RB][CP]) {
        if (z == split) {                                 // workgroup-uniform
#pragma unroll
          for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int cc = 0; cc < CP; ++cc) p[i][cc] = val_of(b0 + i, cc);
          return;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const int m = row_of(b0 + i);
          const bool ok = m < M && col_ok;
          const int off = ok ? (int)((((size_t)z * M + m) * N + ncol) * 4) : 0;
          if constexpr (CP == 4) {
            const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, kAuxSystem);
            p[i][0] = __uint_as_float(q.x); p[i][1] = __uint_as_float(q.y);
            p[i][2] = __uint_as_float(q.z); p[i][3] = __uint_as_float(q.w);
          } else {
            const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, kAuxSystem);
            p[i][0] = __uint_as_float(q.x); p[i][1] = __uint_as_float(q.y);
          }
        }
      };
      auto add = [&](const float (&p)[RB][CP]) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int cc = 0; cc < CP; ++cc) v[i][cc] += p[i][cc];
      };
      float pa[RB][CP], pb[RB][CP];
      take(0, pa);
      for (int z = 0; z < ksplit; z += 2) {
        if (z + 1 < ksplit) take(z + 1, pb);
        add(pa);
        if (z + 2 < ksplit) take(z + 2, pa);
        if (z + 1 < ksplit) add(pb);
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int m = row_of(b0 + i);
      if (m >= M || !col_ok) continue;
      uint32_t o[CP / 2];
#pragma unroll
      for (int cc = 0; cc < CP; cc += 2)
        o[cc / 2] = (uint32_t)float_to_f16_bits(v[i][cc]) | ((uint32_t)float_to_f16_bits(v[i][cc + 1]) << 16);
      if constexpr (CP == 4) *reinterpret_cast<uint2*>(c + (size_t)m * N + ncol) = make_uint2(o[0], o[1]);
      else *reinterpret_cast<uint32_t*>(c + (size_t)m * N + ncol) = o[0];
    }
  }
  if (ksplit > 1 && threadIdx.x == 0)                    // ready for the next call
    __hip_atomic_store(&counters[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Variant "k across waves": grid (column tiles of 32*CP, row tiles, k splits); the 8 waves of a
// workgroup split its k range and add up in LDS.
template <int MT, int CP, int PF>
__global__ void __launch_bounds__(kGemmWaves * 64)
gptq_gemm_kwaves_kernel(uint16_t* __restrict__ c, const uint16_t* __restrict__ a, const uint32_t* __restrict__ qweight,
                        const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ scales, int M, int N, int K,
                        int group_size, int ksplit, int row_tiles, int zoff, float* __restrict__ partials,
                        int32_t* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5;
  // row tiles of one column tile are neighbours in launch order: they stream the same weights
  const int col_tile = blockIdx.x / row_tiles, row_tile = blockIdx.x % row_tiles;
  const int ncol = col_tile * 32 * CP + CP * (lane & 31);
  const bool col_ok = ncol < N;
  const int m0 = row_tile * 32 * MT;
  const int pairs = K / 32;
  const int nw = ksplit * kGemmWaves, wg = blockIdx.z * kGemmWaves + wave;
  const int s_begin = (int)((int64_t)pairs * wg / nw), s_end = (int)((int64_t)pairs * (wg + 1) / nw);

  f32x16 acc[MT][CP];
  gptq_stream<MT, CP, PF>(acc, a, qweight, qzeros, scales, M, N, K, group_size, m0, col_ok ? ncol : 0, s_begin, s_end, zoff);

  // two passes of 8 accumulator rows; wave w then owns accumulator row 8*pass + w
  float vals[2 * MT][CP];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int cc = 0; cc < CP; ++cc)
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8)
          red[(((wave * MT + t) * CP + cc) * 8 + r8) * 64 + lane] = acc[t][cc][8 * pass + r8];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int cc = 0; cc < CP; ++cc) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kGemmWaves; ++w) v += red[(((w * MT + t) * CP + cc) * 8 + wave) * 64 + lane];
        vals[pass * MT + t][cc] = v;
      }
  }
  auto row_of = [&](int i) {                            // i = pass * MT + t, accumulator row r = 8*pass + wave
    const int pass = i / MT, t = i % MT;
    return m0 + 32 * t + 16 * pass + 8 * (wave >> 2) + 4 * half + (wave & 3);
  };
  auto val_of = [&](int i, int cc) { return vals[i][cc]; };
  combine_and_store<CP, 2 * MT>(val_of, row_of, col_ok, ncol, c, M, N, ksplit, partials, counters, blockIdx.x,
                                blockIdx.z, &s_last);
}

// Variant "tiled" (more than 32 rows of A): grid (groups of 8 column tiles, row tiles of 32*MT,
// k splits).  The 8 waves of a workgroup own 8 ADJACENT column tiles over the same k range, so
// every weight word is unpacked exactly once per row tile and reused for all 32*MT rows, and
// the A tile they all need goes through LDS in chunks of 256 k (double buffered, one barrier
// per chunk, the next chunk's global loads in flight during this chunk's math).  k is split across workgroups only.
constexpr int kChunkK = 256;                        // k per A chunk = 16 steps
constexpr int kARow = kChunkK * 2 + 16;             // LDS bytes per A row; the pad keeps 16 rows on 64 distinct banks

template <int MT, int CP, int CHUNK, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
gptq_gemm_tiled_kernel(uint16_t* __restrict__ c, const uint16_t* __restrict__ a, const uint32_t* __restrict__ qweight,
                       const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ scales, int M, int N, int K,
                       int group_size, int ksplit, int col_tiles, int row_tiles, int zoff, float* __restrict__ partials,
                       int32_t* __restrict__ counters) {
  using wvec = typename WordVec<CP>::type;
  using svec = typename WordVec<CP>::scale_t;
  constexpr int PF = 8;                               // weight steps in flight
  constexpr int kARowC = CHUNK * 2 + 16;               // LDS bytes per A row (padded as kARow)
  constexpr int SPC = CHUNK / 16;                   // steps per chunk
  static_assert(SPC % PF == 0, "ring slots must line up with chunks");
  constexpr int BM = 32 * MT;
  constexpr int SEGS = CHUNK / 8;                   // 16-byte pieces per A row of a chunk
  constexpr int PIECES = BM * SEGS;
  constexpr int PPT = (PIECES + WAVES * 64 - 1) / (WAVES * 64);
  extern __shared__ __attribute__((aligned(16))) char a_lds[];   // 2 x BM x kARow
  __shared__ int s_last;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, half = lane >> 5;
  // Workgroup ids go round-robin over the 8 XCDs (each with its own L2).  The row tiles of one
  // column group stream the same weights, so they are made neighbours ON ONE XCD: id -> (xcd =
  // id % 8, slot = id / 8), and the slots of an XCD walk (row tile fastest, column group, k split).
  const int total = col_tiles * row_tiles * ksplit;
  const int per_xcd = (total + 7) / 8;
  const int work = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (work >= total) return;                          // workgroup-uniform (the grid is padded to 8 * per_xcd)
  const int split = work % ksplit;                    // the k splits of a tile run side by side on one XCD: the last
  const int row_tile = (work / ksplit) % row_tiles;   // to arrive finds the others' partial tiles in that XCD's L2
  const int col_tile = work / (ksplit * row_tiles);
  const int ncol = (col_tile * WAVES + wave) * 32 * CP + CP * j;
  const bool col_ok = ncol < N;
  const int ncl = col_ok ? ncol : 0;
  const int m0 = row_tile * BM;
  const int chunks = K / CHUNK;
  const int c_begin = (int)((int64_t)chunks * split / ksplit), c_end = (int)((int64_t)chunks * (split + 1) / ksplit);
  const int s_begin = SPC * c_begin, s_end = SPC * c_end;
  const int spg = group_size / 16;

  f32x16 acc[MT][CP];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int cc = 0; cc < CP; ++cc)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][cc][r] = 0.f;

  // ---- A staging: piece p = (row, 16-byte segment)
  u32x4 a_stage[PPT];                                 // (a native vector: arrays of HIP's uint4 struct, assigned
                                                      //  under a condition, are kept in scratch memory by the compiler)
  // (compile-time slot indices and no early exit from the loop body: the staging registers must
  // not end up addressed through memory)
  auto a_fetch = [&](int chunk) {
    static_for<PPT>([&](auto Q) {
      constexpr int qi = decltype(Q)::value;
      const int p = tid + qi * WAVES * 64;
      if (PIECES % (WAVES * 64) == 0 || p < PIECES) {
        const int row = p / SEGS, seg = p % SEGS;
        a_stage[qi] = *reinterpret_cast<const u32x4*>(a + (size_t)min(m0 + row, M - 1) * K + chunk * CHUNK + seg * 8);
      }
    });
  };
  auto a_commit = [&](int buf) {
    static_for<PPT>([&](auto Q) {
      constexpr int qi = decltype(Q)::value;
      const int p = tid + qi * WAVES * 64;
      if (PIECES % (WAVES * 64) == 0 || p < PIECES) {
        const int row = p / SEGS, seg = p % SEGS;
        *reinterpret_cast<u32x4*>(a_lds + buf * BM * kARowC + row * kARowC + seg * 16) = a_stage[qi];
      }
    });
  };

  // ---- weight ring (as in gptq_stream)
  const uint32_t* wp = qweight + (size_t)half * N + ncl;
  const int zshift = 4 * (ncl & 7);
  wvec wq[PF];
  uint32_t zq[PF];
  svec sq[PF];
  GptqCol col[CP];
  int i_group = s_begin / spg, i_boundary = (i_group + 1) * spg, c_boundary = i_boundary;
  auto issue = [&](auto SLOT, int s) {
    constexpr int slot = decltype(SLOT)::value;
    if (s >= s_end) return;
    wq[slot] = *reinterpret_cast<const wvec*>(wp + (size_t)(2 * s) * N);
    if (s == s_begin || s == i_boundary) {
      if (s == i_boundary) { ++i_group; i_boundary += spg; }
      zq[slot] = qzeros[(size_t)i_group * (N / 8) + ncl / 8];
      sq[slot] = *reinterpret_cast<const svec*>(scales + (size_t)i_group * N + ncl);
    }
  };
  auto consume = [&](auto SLOT, int s, const char* a_buf, int step_in_chunk) {
    constexpr int slot = decltype(SLOT)::value;
    if (s == s_begin || s == c_boundary) {
      if (s == c_boundary) c_boundary += spg;
      const uint32_t z = zq[slot] >> zshift;
      if constexpr (CP == 4) {
        col[0] = gptq_col((z & 0xfu) + zoff, (uint16_t)(sq[slot].x & 0xffffu));
        col[1] = gptq_col(((z >> 4) & 0xfu) + zoff, (uint16_t)(sq[slot].x >> 16));
        col[2] = gptq_col(((z >> 8) & 0xfu) + zoff, (uint16_t)(sq[slot].y & 0xffffu));
        col[3] = gptq_col(((z >> 12) & 0xfu) + zoff, (uint16_t)(sq[slot].y >> 16));
      } else {
        col[0] = gptq_col((z & 0xfu) + zoff, (uint16_t)(sq[slot] & 0xffffu));
        col[1] = gptq_col(((z >> 4) & 0xfu) + zoff, (uint16_t)(sq[slot] >> 16));
      }
    }
    f16x8 af[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
      af[t] = *reinterpret_cast<const f16x8*>(a_buf + (32 * t + j) * kARowC + step_in_chunk * 32 + half * 16);
    uint32_t w[CP];
    if constexpr (CP == 4) { w[0] = wq[slot].x; w[1] = wq[slot].y; w[2] = wq[slot].z; w[3] = wq[slot].w; }
    else { w[0] = wq[slot].x; w[1] = wq[slot].y; }
#pragma unroll
    for (int cc = 0; cc < CP; ++cc) {
      const f16x8 b = gptq_dequant8(w[cc], col[cc]);
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t][cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], b, acc[t][cc], 0, 0, 0);
    }
  };

  static_for<PF>([&](auto SLOT) { issue(SLOT, s_begin + decltype(SLOT)::value); });
  if (c_begin < c_end) a_fetch(c_begin);
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const int par = (chunk - c_begin) & 1;
    a_commit(par);
    __syncthreads();
    if (chunk + 1 < c_end) a_fetch(chunk + 1);
    const char* a_buf = a_lds + par * BM * kARowC;
    static_for<SPC>([&](auto STEP) {                    // SPC is a multiple of PF: ring slots are compile-time
      constexpr int st = decltype(STEP)::value;
      const int s = SPC * chunk + st;
      consume(std::integral_constant<int, st % PF>{}, s, a_buf, st);
      issue(std::integral_constant<int, st % PF>{}, s + PF);
    });
  }

  auto row_of = [&](int i) {
    const int t = i / 16, r = i % 16;
    return m0 + 32 * t + 8 * (r >> 2) + 4 * half + (r & 3);
  };
  auto val_of = [&](int i, int cc) { return acc[i / 16][cc][i % 16]; };
  combine_and_store<CP, 16 * MT>(val_of, row_of, col_ok, ncol, c, M, N, ksplit, partials, counters,
                                 row_tile * col_tiles + col_tile, split, &s_last);
}

// Variant "128-row tiles" (rows of A in blocks of 128, groups that are multiples of the wave's k part): a workgroup
// owns 128 rows x (8 / KQ) * 64 columns over its k range.  Its 8 waves are 8 / KQ column groups of 64 columns x KQ
// k parts: in every chunk of KQ * SPW * 16 k of A (three buffers in LDS, one barrier per chunk) a wave multiplies its
// SPW steps of 16 k, so one unpacked weight word feeds four matrix instructions, and the k parts meet once, through
// LDS, at the end (the wave of k part q ends up with row blocks q * 4 / KQ ..).  The A pieces of chunk n+2 are
// requested while chunk n is multiplied and written to LDS a whole chunk later; the weights run 8 steps ahead.  The
// loop over a chunk has no branch in it: prefetches past the end of the k range re-read the last chunk instead.
constexpr int kT128ARowPad = 16;                     // the pad keeps 16 rows on distinct bank groups
template <int KQ, int SPW> constexpr int t128_chunk() { return KQ * SPW * 16; }
template <int MT, int KQ, int SPW, int NBUF> constexpr int t128_lds() {
  return NBUF * 32 * MT * (t128_chunk<KQ, SPW>() * 2 + kT128ARowPad);
}

template <int MT, int KQ, int CG, int SPW, int NBUF>
__global__ void __launch_bounds__(KQ * CG * 64)
gptq_gemm_tile128_kernel(uint16_t* __restrict__ c, const uint16_t* __restrict__ a, const uint32_t* __restrict__ qweight,
                         const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ scales, int M, int N, int K,
                         int group_size, int ksplit, int col_tiles, int row_tiles, int zoff,
                         float* __restrict__ partials, int32_t* __restrict__ counters) {
  constexpr int CP = 2;                               // columns per lane
  constexpr int NT = KQ * CG * 64;                    // threads
  constexpr int BM = 32 * MT;                         // rows of A per workgroup
  constexpr int BPO = MT / KQ;                        // row blocks a wave keeps at the end
  constexpr int RD = 8 / SPW;                         // chunks of weights in flight (8 steps)
  constexpr int CHUNK = t128_chunk<KQ, SPW>();
  constexpr int AROW = CHUNK * 2 + kT128ARowPad;      // LDS bytes per A row
  constexpr int ABUF = BM * AROW;
  constexpr int SEGS = CHUNK / 8;                     // 16-byte pieces per A row
  constexpr int PPT = BM * SEGS / NT;                 // pieces per thread and chunk
  constexpr int PH = PPT / 2;                         // moved in two halves, at the top and in the middle of a chunk
  constexpr bool kAllK = KQ > MT;                     // more k parts than row blocks: the parts meet as in the k-waves kernel
  static_assert(PPT % 2 == 0 && (BPO >= 1 || (kAllK && CG == 1 && KQ == 8)), "tile shape");
  static_assert(kAllK || KQ * CG * 32 * 64 * 4 <= NBUF * ABUF,
                "the A buffers must hold one 32-register block per wave at the end");
  static_assert(!kAllK || 8 * MT * CP * 8 * 64 * 4 <= NBUF * ABUF,
                "the A buffers must hold 8 accumulator rows of every wave at the end");
  extern __shared__ __attribute__((aligned(16))) char a_lds[];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % CG, kq = wave / CG;
  const int j = lane & 31, half = lane >> 5;
  // workgroup order as in the tiled kernel: the k splits and row tiles of a column tile share an XCD
  const int total = col_tiles * row_tiles * ksplit;
  const int per_xcd = (total + 7) / 8;
  const int work = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (work >= total) return;
  const int split = work % ksplit;
  const int row_tile = (work / ksplit) % row_tiles;
  const int col_tile = work / (ksplit * row_tiles);
  const int ncol = (col_tile * CG + cg) * 64 + CP * j;
  const bool col_ok = ncol < N;
  const int ncl = col_ok ? ncol : 0;
  const int m0 = row_tile * BM;
  const int chunks = K / CHUNK;
  const int c_begin = (int)((int64_t)chunks * split / ksplit), c_end = (int)((int64_t)chunks * (split + 1) / ksplit);

  f32x16 acc[MT][CP];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int cc = 0; cc < CP; ++cc)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][cc][r] = 0.f;
  const DequantRegs dq = dequant_regs();

  // ---- A staging: 128 rows x SEGS pieces of 16 bytes per chunk, PPT per thread in two halves
  constexpr int SETS = NBUF == 3 ? 2 : 1;             // register sets for pieces in flight
  u32x4 a_stage[SETS][PH];
  auto a_fetch = [&](auto HALF, int chunk) {
    constexpr int hf = decltype(HALF)::value;
    const int ch = min(chunk, c_end - 1);
    static_for<PH>([&](auto Q) {
      constexpr int qi = decltype(Q)::value;
      const int p = tid + (hf * PH + qi) * NT;
      const int row = p / SEGS, seg = p % SEGS;
      a_stage[hf % SETS][qi] =
          *reinterpret_cast<const u32x4*>(a + (size_t)min(m0 + row, M - 1) * K + ch * CHUNK + seg * 8);
    });
  };
  auto a_commit = [&](auto HALF, int buf_off) {
    constexpr int hf = decltype(HALF)::value;
    static_for<PH>([&](auto Q) {
      constexpr int qi = decltype(Q)::value;
      const int p = tid + (hf * PH + qi) * NT;
      const int row = p / SEGS, seg = p % SEGS;
      *reinterpret_cast<u32x4*>(a_lds + buf_off + row * AROW + seg * 16) = a_stage[hf % SETS][qi];
    });
  };

  // ---- weights: RD chunks (8 steps of this wave) in flight, slot = chunk index mod RD
  const uint32_t* wp = qweight + (size_t)half * N + ncl;
  const int zshift = 4 * (ncl & 7);
  uint2 wq[RD][SPW];
  uint32_t zq[RD], sq[RD];
  auto w_load = [&](auto SLOT, auto STEP, int chunk) {
    constexpr int slot = decltype(SLOT)::value, i = decltype(STEP)::value;
    const int s = (min(chunk, c_end - 1) * KQ + kq) * SPW + i;
    wq[slot][i] = *reinterpret_cast<const uint2*>(wp + (size_t)(2 * s) * N);
  };
  auto g_load = [&](auto SLOT, int chunk) {
    constexpr int slot = decltype(SLOT)::value;
    const int g = (min(chunk, c_end - 1) * KQ + kq) * (SPW * 16) / group_size;
    zq[slot] = qzeros[(size_t)g * (N / 8) + ncl / 8];
    sq[slot] = *reinterpret_cast<const uint32_t*>(scales + (size_t)g * N + ncl);
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  // One chunk = 2 * SPW half steps (step i, column c of the lane's two).  The instruction order is pinned: the weight
  // word of half step h+1 is unpacked, and the A fragments of step i+1 are read, between the four matrix instructions
  // of half step h, so the matrix pipe never waits for the vector pipe inside a chunk.
  auto body = [&](auto SLOT, int chunk, int buf_cur, int buf_next) {
    constexpr int slot = decltype(SLOT)::value;
    // Three buffers: the pieces of chunk + 2 are requested while those of chunk + 1, requested a chunk ago, go into
    // buf_next (nobody reads it any more after this barrier).  Two buffers: a thread holds one half of a chunk at a
    // time, so a half is written half a chunk after it was requested.
    if constexpr (MT > 4) {                              // 256 accumulator registers: keep them in the AGPR half
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int cc = 0; cc < CP; ++cc) asm volatile("" : "+a"(acc[t][cc]));
    }
    if constexpr (NBUF == 2) a_commit(H1{}, buf_cur * ABUF);
    __syncthreads();
    if constexpr (NBUF == 3) a_commit(H0{}, buf_next * ABUF);
    a_fetch(H0{}, chunk + (NBUF == 3 ? 2 : 1));
    const uint32_t z = zq[slot] >> zshift;
    GptqCol col[CP];
    col[0] = gptq_col((z & 0xfu) + zoff, (uint16_t)(sq[slot] & 0xffffu));
    col[1] = gptq_col(((z >> 4) & 0xfu) + zoff, (uint16_t)(sq[slot] >> 16));
    g_load(SLOT, chunk + RD);
    const char* a_buf = a_lds + buf_cur * ABUF + j * AROW + (kq * SPW) * 32 + half * 16;
    f16x8 af[2][MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) af[0][t] = *reinterpret_cast<const f16x8*>(a_buf + 32 * t * AROW);
    f16x8 b_next = gptq_dequant8(wq[slot][0].x, col[0], dq);
    __builtin_amdgcn_sched_barrier(0);
    static_for<2 * SPW>([&](auto H) {
      constexpr int h = decltype(H)::value, i = h / 2, cc = h % 2;
      const f16x8 b_cur = b_next;
      if constexpr (h + 1 < 2 * SPW) {
        constexpr int i2 = (h + 1) / 2, c2 = (h + 1) % 2;
        b_next = gptq_dequant8(c2 ? wq[slot][i2].y : wq[slot][i2].x, col[c2], dq);
      }
      if constexpr (cc == 0 && i + 1 < SPW) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
          af[(i + 1) & 1][t] = *reinterpret_cast<const f16x8*>(a_buf + 32 * t * AROW + (i + 1) * 32);
      }
      if constexpr (cc == 1) w_load(SLOT, std::integral_constant<int, i>{}, chunk + RD);
#pragma unroll
      for (int t = 0; t < MT; ++t)
        acc[t][cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i & 1][t], b_cur, acc[t][cc], 0, 0, 0);
      // 0x8 matrix, 0x2 vector ALU, 0x100 LDS read, 0x20 memory read
      constexpr int VPM = MT >= 8 ? 2 : 4;              // vector instructions placed after each matrix instruction
      if constexpr (cc == 0) {
        static_for<MT>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
      } else {
        static_for<MT>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (h == SPW - 1) {                     // mid chunk: the second half of chunk + 1 goes in, the second
        if constexpr (NBUF == 3) a_commit(H1{}, buf_next * ABUF);
        else a_commit(H0{}, buf_next * ABUF);
        a_fetch(H1{}, chunk + (NBUF == 3 ? 2 : 1));
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };

  if (c_begin < c_end) {
    static_for<RD>([&](auto SLOT) {
      static_for<SPW>([&](auto STEP) { w_load(SLOT, STEP, c_begin + decltype(SLOT)::value); });
      g_load(SLOT, c_begin + decltype(SLOT)::value);
    });
    if constexpr (NBUF == 3) {
      a_fetch(H0{}, c_begin);
      a_fetch(H1{}, c_begin);
      a_commit(H0{}, 0);
      a_commit(H1{}, 0);
      a_fetch(H0{}, c_begin + 1);
      a_fetch(H1{}, c_begin + 1);
    } else {
      a_fetch(H0{}, c_begin);
      a_commit(H0{}, 0);
      a_fetch(H1{}, c_begin);
    }
    int buf = 0;
    for (int chunk = c_begin; chunk < c_end; chunk += RD) {
      static_for<RD>([&](auto SLOT) {
        constexpr int u = decltype(SLOT)::value;
        if (u == 0 || chunk + u < c_end) {
          const int nxt = buf == NBUF - 1 ? 0 : buf + 1;
          body(SLOT, chunk + u, buf, nxt);
          buf = nxt;
        }
      });
    }
  }

  if constexpr (kAllK) {
    // ---- 8 k parts, one column group: the parts meet as in the k-waves kernel -- two passes of 8 accumulator rows
    // through LDS, after which wave w owns accumulator row 8 * pass + w of every row block.
    float* red = reinterpret_cast<float*>(a_lds);
    float vals[2 * MT][CP];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int cc = 0; cc < CP; ++cc)
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8)
            red[(((wave * MT + t) * CP + cc) * 8 + r8) * 64 + lane] = acc[t][cc][8 * pass + r8];
      __syncthreads();
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int cc = 0; cc < CP; ++cc) {
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < 8; ++w) v += red[(((w * MT + t) * CP + cc) * 8 + wave) * 64 + lane];
          vals[pass * MT + t][cc] = v;
        }
    }
    auto row_of = [&](int i) {                            // i = pass * MT + t, accumulator row r = 8 * pass + wave
      const int pass = i / MT, t = i % MT;
      return m0 + 32 * t + 16 * pass + 8 * (wave >> 2) + 4 * half + (wave & 3);
    };
    auto val_of = [&](int i, int cc) { return vals[i][cc]; };
    combine_and_store<CP, 2 * MT>(val_of, row_of, col_ok, ncol, c, M, N, ksplit, partials, counters,
                                  row_tile * col_tiles + col_tile, split, &s_last);
  } else {
  // ---- the k parts meet: the wave of k part q keeps row blocks q * BPO .. + BPO - 1 and is sent the other parts'
  // sums of them through LDS, one block per pass.
  // (selects between VALUES of fixed accumulator elements: a branch or a conditional expression between two
  // elements becomes a select between their addresses, which keeps the whole accumulator array in scratch memory)
  float* red = reinterpret_cast<float*>(a_lds);
  auto block_of = [&](int q, auto TT, int cc, int r) {  // acc[q * BPO + tt][cc][r] for a wave-uniform runtime q
    constexpr int tt = decltype(TT)::value;
    float v = acc[tt][cc][r];
#pragma unroll
    for (int o = 1; o < KQ; ++o) { const float w = acc[o * BPO + tt][cc][r]; v = q == o ? w : v; }
    return v;
  };
  static_for<(KQ - 1) * BPO>([&](auto P) {
    constexpr int d = 1 + decltype(P)::value / BPO, tt = decltype(P)::value % BPO;
    const int to = (kq + d) % KQ, from = (kq + KQ - d) % KQ;
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < CP; ++cc)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        red[((wave * 32 + cc * 16 + r) << 6) + lane] = block_of(to, std::integral_constant<int, tt>{}, cc, r);
    __syncthreads();
    const int src_wave = from * CG + cg;
#pragma unroll
    for (int cc = 0; cc < CP; ++cc)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = red[((src_wave * 32 + cc * 16 + r) << 6) + lane];
#pragma unroll
        for (int o = 0; o < KQ; ++o) acc[o * BPO + tt][cc][r] += kq == o ? v : 0.f;   // only the kept block changes
      }
  });
  auto row_of = [&](int i) {
    const int t = i / 16, r = i % 16;
    return m0 + 32 * (kq * BPO + t) + 8 * (r >> 2) + 4 * half + (r & 3);
  };
  auto val_of = [&](int i, int cc) {                    // acc[kq * BPO + i / 16][cc][i % 16], by value selects
    float v = acc[i / 16][cc][i % 16];
#pragma unroll
    for (int o = 1; o < KQ; ++o) { const float w = acc[o * BPO + i / 16][cc][i % 16]; v = kq == o ? w : v; }
    return v;
  };
  combine_and_store<CP, 16 * BPO>(val_of, row_of, col_ok, ncol, c, M, N, ksplit, partials, counters,
                                  row_tile * col_tiles + col_tile, split, &s_last);
  }
}

// ------------------------------------------------------------------ GPTQ, plain (unshuffled) weights + g_idx
// The slow, always-correct path of gemm_half_q_half_alt (q_gemm.cu:1288-1383): rows may belong
// to any group.  64 columns x 4 k-slices per workgroup, up to 8 rows of A per pass.
__global__ void __launch_bounds__(256)
gptq_alt_gemm_kernel(uint16_t* __restrict__ c, const uint16_t* __restrict__ a, const uint32_t* __restrict__ qweight,
                     const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ scales,
                     const int32_t* __restrict__ g_idx, int M, int N, int K) {
  __shared__ float red[4][8][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const int m0 = blockIdx.y * 8;
  const int rows = min(8, M - m0);
  float acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = 0.f;
  if (n < N) {
    const int zsh = 4 * (n & 7);
    for (int r = ty; r < K / 8; r += 4) {
      const uint32_t w = qweight[(size_t)r * N + n];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 8 * r + i;
        const int g = g_idx[k];
        const int zero = (int)((qzeros[(size_t)g * (N / 8) + n / 8] >> zsh) & 0xfu) + 1;
        const _Float16 s = __builtin_bit_cast(_Float16, scales[(size_t)g * N + n]);
        const float wv = (float)((_Float16)((int)((w >> (4 * i)) & 0xfu) - zero) * s);
        for (int m = 0; m < rows; ++m) acc[m] += f16_bits_to_float(a[(size_t)(m0 + m) * K + k]) * wv;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) red[ty][m][tx] = acc[m];
  __syncthreads();
  if (n < N)
    for (int m = ty; m < rows; m += 4) {
      const float v = red[0][m][tx] + red[1][m][tx] + red[2][m][tx] + red[3][m][tx];
      c[(size_t)(m0 + m) * N + n] = float_to_f16_bits(v);
    }
}

// ------------------------------------------------------------------ GPTQ helpers
__global__ void gptq_shuffle_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src,
                                    const int32_t* __restrict__ q_perm, int word_rows, int N) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)word_rows * N) return;
  const int r = (int)(idx / N), n = (int)(idx - (int64_t)r * N);
  uint32_t w;
  if (q_perm) {                                       // gather the 8 source rows of this word first
    w = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int srow = q_perm[8 * r + i];
      const uint32_t sw = src[(size_t)(srow >> 3) * N + n];
      w |= ((sw >> (4 * (srow & 7))) & 0xfu) << (4 * i);
    }
  } else {
    w = src[idx];
  }
  dst[idx] = shuffle_nibbles(w);
}

__global__ void permute_columns_kernel(uint16_t* __restrict__ dst, const uint16_t* __restrict__ src,
                                       const int32_t* __restrict__ perm, int M, int K) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * K) return;
  const int m = (int)(idx / K), k = (int)(idx - (int64_t)m * K);
  dst[idx] = src[(size_t)m * K + perm[k]];
}

// one thread: one word row x 4 columns -> 8 rows x 4 fp16
__global__ void gptq_dequant_kernel(uint16_t* __restrict__ out, const uint32_t* __restrict__ qweight,
                                    const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ scales,
                                    const int32_t* __restrict__ g_idx, int K, int N, int group_size, int shuffled, int zoff) {
  const int n4 = N / 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)(K / 8) * n4) return;
  const int r = (int)(idx / n4), n = 4 * (int)(idx - (int64_t)r * n4);
  const uint4 w4 = *reinterpret_cast<const uint4*>(qweight + (size_t)r * N + n);
  const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
  if (shuffled) {
    const int g = 8 * r / group_size;
    const uint32_t z = qzeros[(size_t)g * (N / 8) + n / 8] >> (4 * (n & 7));
    const uint2 s = *reinterpret_cast<const uint2*>(scales + (size_t)g * N + n);
    const uint16_t sb[4] = {(uint16_t)(s.x & 0xffffu), (uint16_t)(s.x >> 16), (uint16_t)(s.y & 0xffffu),
                            (uint16_t)(s.y >> 16)};
    union { f16x8 f; uint32_t u[4]; } d[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) d[cc].f = gptq_dequant8(w[cc], gptq_col(((z >> (4 * cc)) & 0xfu) + zoff, sb[cc]));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = g_idx ? g_idx[8 * r + i] : 8 * r + i;     // g_idx holds the row permutation here
      const int sh = 16 * (i & 1);
      uint2 o;
      o.x = ((d[0].u[i >> 1] >> sh) & 0xffffu) | (((d[1].u[i >> 1] >> sh) & 0xffffu) << 16);
      o.y = ((d[2].u[i >> 1] >> sh) & 0xffffu) | (((d[3].u[i >> 1] >> sh) & 0xffffu) << 16);
      *reinterpret_cast<uint2*>(out + (size_t)row * N + n) = o;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 8 * r + i;
      const int g = g_idx ? g_idx[k] : k / group_size;
      const uint32_t z = qzeros[(size_t)g * (N / 8) + n / 8] >> (4 * (n & 7));
      uint16_t o[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int zero = (int)((z >> (4 * cc)) & 0xfu) + 1;
        const _Float16 s = __builtin_bit_cast(_Float16, scales[(size_t)g * N + n + cc]);
        o[cc] = __builtin_bit_cast(uint16_t, (_Float16)((_Float16)((int)((w[cc] >> (4 * i)) & 0xfu) - zero) * s));
      }
      uint2 ov;
      ov.x = o[0] | ((uint32_t)o[1] << 16);
      ov.y = o[2] | ((uint32_t)o[3] << 16);
      *reinterpret_cast<uint2*>(out + (size_t)k * N + n) = ov;
    }
  }
}

// ------------------------------------------------------------------ GPTQ, 8-bit weights
// csrc/quantization/gptq/qdq_8.cuh and the 8-bit kernels of q_gemm.cu (:601-735 fused GEMM, :810-918 reconstruction,
// :1989-2029 row gather): a word holds 4 consecutive k of one column, byte i = row 4r + i, in the same order before
// and after "shuffling" (shuffle_8bit_4 is empty); qzeros [groups, N/4], byte n % 4 = zero - 1.  No configuration
// of the benchmark uses 8-bit weights: the kernels below are plain ones (no matrix cores), and the Python operator
// reconstructs and calls a dense GEMM above 24 rows as the reference does (MAX_Q_GEMM_ROWS_8BIT, q_gemm.cu:1707).
__global__ void gptq8_gather_rows_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src,
                                         const int32_t* __restrict__ q_perm, int word_rows, int N) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)word_rows * N) return;
  const int r = (int)(idx / N), n = (int)(idx - (int64_t)r * N);
  uint32_t w = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int srow = q_perm[4 * r + i];
    w |= ((src[(size_t)(srow >> 2) * N + n] >> (8 * (srow & 3))) & 0xffu) << (8 * i);
  }
  dst[idx] = w;
}

// W[k][n] of row k of the (possibly gathered) weight: exact integer difference, one fp16 rounding for the product
__device__ __forceinline__ _Float16 gptq8_weight(uint32_t word, int i, const uint32_t* __restrict__ qzeros,
                                                 const uint16_t* __restrict__ scales, int g, int n, int N) {
  const int zero = (int)((qzeros[(size_t)g * (N / 4) + n / 4] >> (8 * (n & 3))) & 0xffu) + 1;
  const _Float16 s = __builtin_bit_cast(_Float16, scales[(size_t)g * N + n]);
  return (_Float16)((_Float16)((int)((word >> (8 * i)) & 0xffu) - zero) * s);
}

// `rows`: exllama form -- groups are consecutive rows, g_idx (may be NULL) is the row permutation: weight row k is
// input column / output row g_idx[k]; plain form -- g_idx[k] is the group of row k.
__global__ void gptq8_dequant_kernel(uint16_t* __restrict__ out, const uint32_t* __restrict__ qweight,
                                     const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ scales,
                                     const int32_t* __restrict__ g_idx, int K, int N, int group_size, int exllama) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)(K / 4) * N) return;
  const int r = (int)(idx / N), n = (int)(idx - (int64_t)r * N);
  const uint32_t w = qweight[idx];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = 4 * r + i;
    const int g = exllama ? k / group_size : g_idx[k];
    const int row = (exllama && g_idx) ? g_idx[k] : k;
    out[(size_t)row * N + n] = __builtin_bit_cast(uint16_t, gptq8_weight(w, i, qzeros, scales, g, n, N));
  }
}

// 64 columns x 4 k-slices per workgroup, up to 8 rows of A per workgroup (grid.y walks the rows)
__global__ void __launch_bounds__(256)
gptq8_gemm_kernel(uint16_t* __restrict__ c, const uint16_t* __restrict__ a, const uint32_t* __restrict__ qweight,
                  const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ scales,
                  const int32_t* __restrict__ g_idx, int M, int N, int K, int group_size, int exllama) {
  __shared__ float red[4][8][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const int m0 = blockIdx.y * 8;
  const int rows = min(8, M - m0);
  float acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = 0.f;
  if (n < N) {
    for (int r = ty; r < K / 4; r += 4) {
      const uint32_t w = qweight[(size_t)r * N + n];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 4 * r + i;
        const int g = exllama ? k / group_size : g_idx[k];
        const int col = (exllama && g_idx) ? g_idx[k] : k;
        const float wv = (float)gptq8_weight(w, i, qzeros, scales, g, n, N);
        for (int m = 0; m < rows; ++m) acc[m] += f16_bits_to_float(a[(size_t)(m0 + m) * K + col]) * wv;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) red[ty][m][tx] = acc[m];
  __syncthreads();
  if (n < N)
    for (int m = ty; m < rows; m += 4) {
      const float v = red[0][m][tx] + red[1][m][tx] + red[2][m][tx] + red[3][m][tx];
      c[(size_t)(m0 + m) * N + n] = float_to_f16_bits(v);
    }
}

// ------------------------------------------------------------------ AWQ
// A word holds 8 output columns of one input row; masking pairs nibbles (0,4) (1,5) (2,6) (3,7),
// and checkpoints are packed so that those pairs ARE logical columns (0,1) (2,3) (4,5) (6,7)
// (dequantize.cuh:40-56).
struct AwqCols {
  h2 z1[2];    // -(1024 + zero) for column pairs (0,1) and (4,5)
  h2 z16[2];   // -(64 + zero) for column pairs (2,3) and (6,7)
  h2 s[4];     // scales of column pairs (0,1) (2,3) (4,5) (6,7)
};

__device__ __forceinline__ AwqCols awq_cols(uint32_t zw, uint4 sc) {
  const h2 neg16th = splat((_Float16)(-0.0625f));
  AwqCols c;
  c.z1[0] = as_h2(((zw & 0x000f000fu) | kMagic) ^ 0x80008000u);
  c.z16[0] = as_h2((zw & 0x00f000f0u) | kMagic) * neg16th;
  zw >>= 8;
  c.z1[1] = as_h2(((zw & 0x000f000fu) | kMagic) ^ 0x80008000u);
  c.z16[1] = as_h2((zw & 0x00f000f0u) | kMagic) * neg16th;
  c.s[0] = as_h2(sc.x); c.s[1] = as_h2(sc.y); c.s[2] = as_h2(sc.z); c.s[3] = as_h2(sc.w);
  return c;
}

// one word -> 8 fp16 in logical column order
__device__ __forceinline__ uint4 awq_dequant8(uint32_t w, const AwqCols& c) {
  const h2 y16 = splat((_Float16)0.0625f);
  const h2 q0 = as_h2((w & 0x000f000fu) | kMagic);
  const h2 q1 = as_h2((w & 0x00f000f0u) | kMagic);
  w >>= 8;
  const h2 q2 = as_h2((w & 0x000f000fu) | kMagic);
  const h2 q3 = as_h2((w & 0x00f000f0u) | kMagic);
  uint4 r;
  r.x = as_u32((q0 + c.z1[0]) * c.s[0]);
  r.y = as_u32((q1 * y16 + c.z16[0]) * c.s[1]);
  r.z = as_u32((q2 + c.z1[1]) * c.s[2]);
  r.w = as_u32((q3 * y16 + c.z16[1]) * c.s[3]);
  return r;
}

__global__ void awq_dequantize_kernel(uint16_t* __restrict__ out, const uint32_t* __restrict__ qweight,
                                      const uint16_t* __restrict__ scales, const uint32_t* __restrict__ qzeros,
                                      int K, int N, int group_size) {
  const int nw = N / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)K * nw) return;
  const int k = (int)(idx / nw), wc = (int)(idx - (int64_t)k * nw);
  const int g = k / group_size;
  const AwqCols cols = awq_cols(qzeros[(size_t)g * nw + wc],
                                *reinterpret_cast<const uint4*>(scales + (size_t)g * N + 8 * wc));
  *reinterpret_cast<uint4*>(out + (size_t)k * N + 8 * wc) = awq_dequant8(qweight[idx], cols);
}

// AWQ checkpoint -> the k-major layout of the GPTQ kernels (a word = 8 consecutive k of one
// column, even rows in the low half and odd rows in the high half; zeros in plain column order,
// stored without GPTQ's -1).  One thread per output word.
__global__ void awq_to_kmajor_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ qweight, int k8, int N) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)k8 * N) return;
  const int r = (int)(idx / N), n = (int)(idx - (int64_t)r * N);
  const int col = n & 7;
  const int shift = 4 * ((col >> 1) + 4 * (col & 1));          // nibble of logical column col inside an AWQ word
  uint32_t w = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t q = (qweight[(size_t)(8 * r + i) * (N / 8) + n / 8] >> shift) & 0xfu;
    w |= q << (4 * (i >> 1) + 16 * (i & 1));
  }
  out[idx] = w;
}

__global__ void awq_zeros_to_kmajor_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ qzeros, int64_t words) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= words) return;
  const uint32_t zw = qzeros[idx];
  uint32_t w = 0;
#pragma unroll
  for (int col = 0; col < 8; ++col) w |= ((zw >> (4 * ((col >> 1) + 4 * (col & 1)))) & 0xfu) << (4 * col);
  out[idx] = w;
}

// Fused GEMM, up to 32 rows of A per workgroup.  Grid (column tiles of 256 x row tiles, 1, k splits);
// the 8 waves of a workgroup split its k range into runs of 16-k steps and add up in LDS.
//
// A lane (j = lane & 31, half = lane >> 5) loads the 8 words of ONE word column for k rows
// 8*half .. 8*half+7 of a step: an 8 k x 8 column block of nibbles.  The matrix core wants, per
// column, 8 consecutive k in one lane -- a transpose that stays inside the lane: v_perm puts byte
// b of two consecutive rows into the low bytes of the two halves of a dword, where the two masks
// of the magic-number unpacking find the (k, k+1) pair of the byte's low-nibble column and the
// pair of its high-nibble column.  Accumulator c therefore holds columns n0 + 8 j + c, and a
// thread ends up with runs of adjacent columns to store.
constexpr int kAwqWaves = 8;

struct AwqLane {
  h2 zt[8];    // per logical column: -(1024 + zero) (low nibble of its byte) or -(64 + zero) (high nibble)
  h2 s[8];     // per logical column: scale in both halves
};

// logical column of nibble p of a word (dequantize.cuh:40-56 packs 0 2 4 6 1 3 5 7)
__host__ __device__ constexpr int awq_col_of_nibble(int p) { return 2 * (p & 3) + (p >> 2); }

__device__ __forceinline__ AwqLane awq_lane_cols(uint32_t zw, uint4 sc) {
  AwqLane c;
  const uint32_t sw[4] = {sc.x, sc.y, sc.z, sc.w};
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int col = awq_col_of_nibble(p);
    const uint32_t z = (zw >> (4 * p)) & 0xfu;
    if (p & 1) {
      c.zt[col] = splat((_Float16)(-64 - (int)z));
    } else {
      const uint32_t zb = 0xe400u | z;                // -(1024 + z)
      c.zt[col] = as_h2(zb | (zb << 16));
    }
    const uint32_t sb = (sw[col >> 1] >> (16 * (col & 1))) & 0xffffu;
    c.s[col] = as_h2(sb | (sb << 16));
  }
  return c;
}

__global__ void __launch_bounds__(kAwqWaves * 64)
awq_gemm_kernel(uint16_t* __restrict__ c, const uint16_t* __restrict__ a, const uint32_t* __restrict__ qweight,
                const uint16_t* __restrict__ scales, const uint32_t* __restrict__ qzeros, int M, int N, int K,
                int group_size, int ksplit, int row_tiles, float* __restrict__ partials,
                int32_t* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, half = lane >> 5;
  const int nw = N / 8;
  // row tiles of one column tile are neighbours in launch order: they stream the same weights
  const int col_tile = blockIdx.x / row_tiles, row_tile = blockIdx.x % row_tiles;
  const int n0 = col_tile * 256;
  const int wcol = n0 / 8 + j;                        // this lane's word column
  const bool w_ok = wcol < nw;
  const int wcl = w_ok ? wcol : 0;
  const int m0 = row_tile * 32;

  const int steps = K / 16;
  const int nwv = ksplit * kAwqWaves, wg = blockIdx.z * kAwqWaves + wave;
  const int s_begin = (int)((int64_t)steps * wg / nwv), s_end = (int)((int64_t)steps * (wg + 1) / nwv);

  f32x16 acc[8];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;

  const uint16_t* ap = a + (size_t)min(m0 + j, M - 1) * K + 8 * half;
  const uint32_t* wp = qweight + (size_t)(8 * half) * nw + wcl;

  AwqLane cols;
  int cur_group = -1;
  constexpr int PF = 4;                               // 16-k steps in flight per wave
  uint32_t wq[PF][8];
  f16x8 af[PF];
  auto load_step = [&](auto SLOT, int s) {
    constexpr int slot = decltype(SLOT)::value;
    if (s >= s_end) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) wq[slot][i] = wp[(size_t)(16 * s + i) * nw];
    af[slot] = *reinterpret_cast<const f16x8*>(ap + 16 * s);
  };
  auto run_step = [&](auto SLOT, int s) {
    constexpr int slot = decltype(SLOT)::value;
    const int g = 16 * s / group_size;
    if (g != cur_group) {                             // wave-uniform; group data is L2-resident
      cols = awq_lane_cols(qzeros[(size_t)g * nw + wcl], *reinterpret_cast<const uint4*>(scales + (size_t)g * N + 8 * wcl));
      cur_group = g;
    }
    const h2 y16 = splat((_Float16)0.0625f);
    union { uint32_t u[4]; f16x8 f; } b[8];
#pragma unroll
    for (int m = 0; m < 4; ++m) {                     // k rows 2m, 2m+1 of the lane's eight
      const uint32_t w0 = wq[slot][2 * m], w1 = wq[slot][2 * m + 1];
#pragma unroll
      for (int byte = 0; byte < 4; ++byte) {
        const uint32_t u = __builtin_amdgcn_perm(w1, w0, 0x0c000c00u | (uint32_t)byte | ((uint32_t)(4 + byte) << 16));
        const int c_lo = awq_col_of_nibble(2 * byte), c_hi = awq_col_of_nibble(2 * byte + 1);
        const h2 q_lo = as_h2((u & 0x000f000fu) | kMagic);
        const h2 q_hi = as_h2((u & 0x00f000f0u) | kMagic);
        b[c_lo].u[m] = as_u32((q_lo + cols.zt[c_lo]) * cols.s[c_lo]);
        b[c_hi].u[m] = as_u32((q_hi * y16 + cols.zt[c_hi]) * cols.s[c_hi]);
      }
    }
    const f16x8 a_frag = af[slot];
    load_step(SLOT, s + PF);                          // the slot's registers are free again
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_frag, b[cb].f, acc[cb], 0, 0, 0);
  };
  static_for<PF>([&](auto SLOT) { load_step(SLOT, s_begin + decltype(SLOT)::value); });
  for (int s = s_begin; s < s_end; s += PF) {
    static_for<PF>([&](auto SLOT) {
      constexpr int i = decltype(SLOT)::value;
      if (s + i < s_end) run_step(SLOT, s + i);
    });
  }

  // The eight waves' sums meet in LDS in four passes of 4 accumulator rows.  After pass p wave w
  // owns accumulator row 4p + (w & 3) = matrix row m0 + 8p + 4*half + (w & 3), and of the lane's
  // 8 columns the four starting at 4*(w >> 2).
  const int cq = wave >> 2, rw = wave & 3;
  float vals[4][4];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) red[((wave * 8 + cb) * 4 + r4) * 64 + lane] = acc[cb][4 * pass + r4];
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < kAwqWaves; ++w) v += red[((w * 8 + 4 * cq + cc) * 4 + rw) * 64 + lane];
      vals[pass][cc] = v;
    }
  }
  const int ncol = n0 + 8 * j + 4 * cq;
  auto row_of = [&](int pass) { return m0 + 8 * pass + 4 * half + rw; };
  auto val_of = [&](int pass, int cc) { return vals[pass][cc]; };
  combine_and_store<4, 4>(val_of, row_of, w_ok, ncol, c, M, N, ksplit, partials, counters, blockIdx.x, blockIdx.z,
                          &s_last);
}

// how many ways to split k across workgroups: aim at two workgroups per compute unit, keep at
// least `min_units` units of work per wave, and stay inside the workspace
int pick_ksplit(int tiles, int units, int min_units, int M, int N, int64_t workspace_bytes, int waves = 4,
                int target_wgs = 512) {
  if (tiles > kCounterInts) return 1;
  int ks = (target_wgs + tiles / 2) / tiles;
  const int cap = units / (waves * min_units);
  if (ks > cap) ks = cap;
  if (ks > 64) ks = 64;
#ifdef SW_QUANT_EXPERIMENT                                     // tuning builds only: never in the shipped library
  if (const char* env = getenv("SW_QUANT_KSPLIT")) {
    const int forced = atoi(env);
    if (forced >= 1 && forced * waves <= units) ks = forced;
  }
#endif
  const int64_t per_split = (int64_t)M * N * 4;
  const int64_t room = (workspace_bytes - kCounterBytes) / (per_split > 0 ? per_split : 1);
  if (ks > room) ks = (int)room;
  const int64_t addressable = ((int64_t)1 << 31) / (per_split > 0 ? per_split : 1);   // 32-bit buffer offsets
  if (ks > addressable) ks = (int)addressable;
  return ks < 1 ? 1 : ks;
}

}  // namespace
}  // namespace sw

using namespace sw;

extern "C" {

int64_t sw_quant_gemm_workspace_bytes(int M, int N, int K) {
  // counters + up to 8 fp32 partial tiles (fewer splits are used when this is smaller than
  // what the heuristic would like) + a permuted copy of A for act-order GPTQ
  const int64_t partial = (int64_t)M * N * 4;
  return kCounterBytes + 8 * partial + (int64_t)M * K * 2 + 256;
}

int sw_gptq_shuffle(uint32_t* qweight, const int32_t* q_perm, int K, int N, int bit, void* scratch, void* stream) {
  if (bit == 8) {                                     // the 8-bit word order is the checkpoint's; only rows may move
    SW_CHECK_ARG(K > 0 && N > 0 && K % 4 == 0, "gptq_shuffle: bad shape K=%d N=%d", K, N);
    if (q_perm == nullptr) return SW_OK;
    SW_CHECK_ARG(scratch != nullptr, "gptq_shuffle: a row permutation needs a scratch buffer");
    hipStream_t st8 = static_cast<hipStream_t>(stream);
    const int64_t words8 = (int64_t)(K / 4) * N;
    hipLaunchKernelGGL(gptq8_gather_rows_kernel, dim3((unsigned)((words8 + 255) / 256)), dim3(256), 0, st8,
                       static_cast<uint32_t*>(scratch), qweight, q_perm, K / 4, N);
    SW_CHECK_LAUNCH();
    SW_CHECK_HIP(hipMemcpyAsync(qweight, scratch, words8 * 4, hipMemcpyDeviceToDevice, st8));
    return SW_OK;
  }
  SW_CHECK_ARG(bit == 4, "gptq_shuffle: only 4-bit and 8-bit weights are supported (got %d)", bit);
  SW_CHECK_ARG(K > 0 && N > 0 && K % 8 == 0, "gptq_shuffle: bad shape K=%d N=%d", K, N);
  SW_CHECK_ARG(q_perm == nullptr || scratch != nullptr, "gptq_shuffle: a row permutation needs a scratch buffer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t words = (int64_t)(K / 8) * N;
  const int blocks = (int)((words + 255) / 256);
  if (q_perm) {
    uint32_t* tmp = static_cast<uint32_t*>(scratch);
    hipLaunchKernelGGL(gptq_shuffle_kernel, dim3(blocks), dim3(256), 0, st, tmp, qweight, q_perm, K / 8, N);
    SW_CHECK_LAUNCH();
    SW_CHECK_HIP(hipMemcpyAsync(qweight, tmp, words * 4, hipMemcpyDeviceToDevice, st));
  } else {
    hipLaunchKernelGGL(gptq_shuffle_kernel, dim3(blocks), dim3(256), 0, st, qweight, qweight, nullptr, K / 8, N);
    SW_CHECK_LAUNCH();
  }
  return SW_OK;
}

static int kmajor_dequant(void* out, const uint32_t* qweight, const uint32_t* qzeros, const void* scales,
                          const int32_t* g_idx, int K, int N, int groups, int use_exllama, int bit, int zoff,
                          void* stream) {
  if (bit == 8 && zoff == 1) {
    SW_CHECK_ARG(K > 0 && N > 0 && groups > 0 && K % 4 == 0 && N % 4 == 0 && K % groups == 0,
                 "gptq_dequant: bad shape K=%d N=%d groups=%d", K, N, groups);
    SW_CHECK_ARG(use_exllama || g_idx != nullptr, "gptq_dequant: plain weights need g_idx");
    const int64_t words8 = (int64_t)(K / 4) * N;
    hipLaunchKernelGGL(gptq8_dequant_kernel, dim3((unsigned)((words8 + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<uint16_t*>(out), qweight, qzeros,
                       static_cast<const uint16_t*>(scales), g_idx, K, N, K / groups, use_exllama ? 1 : 0);
    SW_CHECK_LAUNCH();
    return SW_OK;
  }
  SW_CHECK_ARG(bit == 4, "gptq_dequant: only 4-bit and 8-bit weights are supported (got %d)", bit);
  SW_CHECK_ARG(K > 0 && N > 0 && groups > 0 && K % 8 == 0 && N % 8 == 0 && K % groups == 0,
               "gptq_dequant: bad shape K=%d N=%d groups=%d", K, N, groups);
  SW_CHECK_ARG(use_exllama || g_idx != nullptr, "gptq_dequant: plain weights need g_idx");
  const int group_size = K / groups;
  SW_CHECK_ARG(!use_exllama || group_size % 8 == 0, "gptq_dequant: group size %d is not a multiple of 8", group_size);
  const int64_t threads = (int64_t)(K / 8) * (N / 4);
  hipLaunchKernelGGL(gptq_dequant_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<uint16_t*>(out), qweight, qzeros,
                     static_cast<const uint16_t*>(scales), g_idx, K, N, group_size, use_exllama ? 1 : 0, zoff);
  SW_CHECK_LAUNCH();
  return SW_OK;
}

int sw_gptq_dequant(void* out, const uint32_t* qweight, const uint32_t* qzeros, const void* scales,
                    const int32_t* g_idx, int K, int N, int groups, int use_exllama, int bit, void* stream) {
  return kmajor_dequant(out, qweight, qzeros, scales, g_idx, K, N, groups, use_exllama, bit, 1, stream);
}

static int kmajor_gemm(void* c, const void* a, const uint32_t* qweight, const uint32_t* qzeros, const void* scales,
                       const int32_t* g_idx, int M, int N, int K, int groups, int use_exllama, int bit, int zoff,
                       void* workspace, int64_t workspace_bytes, void* stream) {
  if (bit == 8 && zoff == 1) {
    SW_CHECK_ARG(M >= 0 && N > 0 && K > 0 && groups > 0 && K % 4 == 0 && N % 4 == 0 && K % groups == 0,
                 "gptq_gemm: bad shape M=%d N=%d K=%d groups=%d", M, N, K, groups);
    SW_CHECK_ARG(use_exllama || g_idx != nullptr, "gptq_gemm: plain (unshuffled) weights need g_idx");
    if (M == 0) return SW_OK;
    hipLaunchKernelGGL(gptq8_gemm_kernel, dim3((N + 63) / 64, (M + 7) / 8), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<uint16_t*>(c), static_cast<const uint16_t*>(a),
                       qweight, qzeros, static_cast<const uint16_t*>(scales), g_idx, M, N, K, K / groups,
                       use_exllama ? 1 : 0);
    SW_CHECK_LAUNCH();
    return SW_OK;
  }
  SW_CHECK_ARG(bit == 4, "gptq_gemm: only 4-bit and 8-bit weights are supported (got %d)", bit);
  SW_CHECK_ARG(M >= 0 && N > 0 && K > 0 && groups > 0 && K % 32 == 0 && N % 8 == 0 && K % groups == 0,
               "gptq_gemm: bad shape M=%d N=%d K=%d groups=%d", M, N, K, groups);
  if (M == 0) return SW_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint16_t* out = static_cast<uint16_t*>(c);
  const uint16_t* ap = static_cast<const uint16_t*>(a);
  const uint16_t* sp = static_cast<const uint16_t*>(scales);
  if (!use_exllama) {
    SW_CHECK_ARG(g_idx != nullptr, "gptq_gemm: plain (unshuffled) weights need g_idx");
    hipLaunchKernelGGL(gptq_alt_gemm_kernel, dim3((N + 63) / 64, (M + 7) / 8), dim3(256), 0, st, out, ap, qweight,
                       qzeros, sp, g_idx, M, N, K);
    SW_CHECK_LAUNCH();
    return SW_OK;
  }
  const int group_size = K / groups;
  SW_CHECK_ARG(group_size % 32 == 0, "gptq_gemm: group size %d is not a multiple of 32", group_size);
  SW_CHECK_ARG(workspace != nullptr && workspace_bytes >= kCounterBytes,
               "gptq_gemm: workspace of at least %d bytes required", kCounterBytes);
  int32_t* counters = static_cast<int32_t*>(workspace);
  char* ws = static_cast<char*>(workspace);
  int64_t avail = workspace_bytes;
  if (g_idx) {                                        // act-order: A's columns follow the shuffled row order
    const int64_t abytes = ((int64_t)M * K * 2 + 255) / 256 * 256;
    SW_CHECK_ARG(workspace_bytes >= kCounterBytes + abytes, "gptq_gemm: workspace too small for the permuted input");
    avail -= abytes;
    uint16_t* a_perm = reinterpret_cast<uint16_t*>(ws + avail);
    const int64_t total = (int64_t)M * K;
    hipLaunchKernelGGL(permute_columns_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a_perm, ap,
                       g_idx, M, K);
    SW_CHECK_LAUNCH();
    ap = a_perm;
  }
  float* partials = reinterpret_cast<float*>(ws + kCounterBytes);
  // Kernel choice, from measurements on the 13B / TP2 matrices (tools/quant_exp.hip):
  //   * up to 32 rows: one row tile per workgroup, k split across its waves, A straight from memory;
  //   * tile kernels (A through LDS; they need k in multiples of 256 and groups in multiples of 64):
  //       33..64 rows    64 x 64 tiles, the 8 waves are 8 k parts of every chunk;
  //       65..128 rows   128 x 64 tiles of the same form -- a 13824-column matrix is 216 of them, so k is not split
  //                      across workgroups and no fp32 partial tile goes through memory; with a short k (<= 3072) and
  //                      few columns two 64-row tiles per column tile, k unsplit, are faster than 128-row tiles whose
  //                      k is split three ways;
  //       129.. rows     128 x 128 tiles (2 column groups x 4 k parts), 128 x 256 when that keeps the launch to one
  //                      round of workgroups (the 27648-column matrix of an unsharded 13B model at 256 rows);
  //   * shapes the tile kernels do not take: k across waves up to 128 rows, the 64-row tiled kernel above.
  int cp = 2, mt = M > 64 ? 4 : M > 32 ? 2 : 1;
  bool tiled = M > 128 && K % kChunkK == 0 && group_size % 16 == 0;
  int waves = 8, chunk = kChunkK, tiled_mt = 2, force_ks = 0;
  int t_variant = 0;
  if (K % 256 == 0 && group_size % 64 == 0) {
    if (M > 128) {
      t_variant = (int64_t)((M + 127) / 128) * ((N + 127) / 128) > 256 ? 2 : 4;
      // a short k and few columns (13B / TP2 o_proj: 22.5 -> 19.5 us at 256 rows): 128 x 64 tiles, k unsplit
      if (K <= 3072 && (int64_t)((M + 127) / 128) * ((N + 63) / 64) <= 256) { t_variant = 8; force_ks = 1; }
    } else if (M > 64) {
      t_variant = 8;
      if (K <= 3072 && 2 * ((N + 63) / 64) <= 256) { t_variant = 68; force_ks = 1; }
    } else if (M > 32) {
      t_variant = 68;
    }
  }
#ifdef SW_QUANT_EXPERIMENT                                     // tuning builds only: never in the shipped library
  if (const char* env = getenv("SW_QUANT_CP")) cp = atoi(env) == 4 ? 4 : 2;
  if (const char* env = getenv("SW_QUANT_MT")) mt = atoi(env);
  if (const char* env = getenv("SW_QUANT_TILED")) tiled = atoi(env) != 0 && K % kChunkK == 0;
  if (const char* env = getenv("SW_QUANT_WAVES")) waves = atoi(env) == 4 ? 4 : 8;
  if (const char* env = getenv("SW_QUANT_CHUNK")) chunk = atoi(env) == 128 ? 128 : 256;
  if (const char* env = getenv("SW_QUANT_TMT")) tiled_mt = atoi(env);
  if (const char* env = getenv("SW_QUANT_KSPLIT")) force_ks = atoi(env);
  // 0: off; 2, 4, 8: 128 rows x 8 waves with 2 / 4 / 8 k parts; 68: 64 rows, 8 k parts; 82, 84: 256 rows x 4 waves
  if (const char* env = getenv("SW_QUANT_T128")) t_variant = atoi(env);
#endif
  if (mt == 3 || mt > 4) mt = 4;
  if (tiled) mt = tiled_mt;
  if (t_variant != 0) {
    // variant -> rows per tile, k parts, column groups of 64, k per chunk
    int t_rows = 128, t_kq = 4, t_cg = 2, chunk_k = 256;
    switch (t_variant) {
      case 4: break;
      case 2: t_kq = 2; t_cg = 4; chunk_k = 128; break;
      case 8: t_kq = 8; t_cg = 1; break;
      case 68: t_rows = 64; t_kq = 8; t_cg = 1; break;
      case 38: t_rows = 32; t_kq = 8; t_cg = 1; break;
      case 82: t_rows = 256; t_kq = 2; t_cg = 2; chunk_k = 128; break;
      case 84: t_rows = 256; t_kq = 4; t_cg = 1; chunk_k = 128; break;
      default: SW_CHECK_ARG(false, "gptq_gemm: no 128-row tile variant %d", t_variant);
    }
    SW_CHECK_ARG(K % 256 == 0 && group_size % 64 == 0,
                 "gptq_gemm: no 128-row tile variant %d for K=%d, group size %d", t_variant, K, group_size);
    const int rt = (M + t_rows - 1) / t_rows, ct = (N + 64 * t_cg - 1) / (64 * t_cg);
    int ks = pick_ksplit(ct * rt, K / chunk_k, 4, M, N, avail, 1, 256);
    if (force_ks > 0 && force_ks <= K / chunk_k) ks = force_ks;
#define SW_GPTQ_T128(MT_, KQ_, CG_, SPW_, NB_)                                                                 \
  do {                                                                                                           \
    auto kern = gptq_gemm_tile128_kernel<MT_, KQ_, CG_, SPW_, NB_>;                                              \
    constexpr int lds_bytes = t128_lds<MT_, KQ_, SPW_, NB_>();                                                   \
    SW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                     lds_bytes));                                                                \
    hipLaunchKernelGGL(kern, dim3((ct * rt * ks + 7) / 8 * 8), dim3(KQ_ * CG_ * 64), lds_bytes, st, out, ap,     \
                       qweight, qzeros, sp, M, N, K, group_size, ks, ct, rt, zoff, partials, counters);          \
  } while (0)
    if (t_variant == 4) SW_GPTQ_T128(4, 4, 2, 4, 2);
    else if (t_variant == 2) SW_GPTQ_T128(4, 2, 4, 4, 3);
    else if (t_variant == 8) SW_GPTQ_T128(4, 8, 1, 2, 2);
    else if (t_variant == 68) SW_GPTQ_T128(2, 8, 1, 2, 2);
#ifdef SW_QUANT_EXPERIMENT
    else if (t_variant == 38) SW_GPTQ_T128(1, 8, 1, 2, 2);    // 32-row tiles: behind the k-across-waves kernel at every shape
    else if (t_variant == 82) SW_GPTQ_T128(8, 2, 2, 4, 2);
    else if (t_variant == 84) SW_GPTQ_T128(8, 4, 1, 2, 2);
#endif
    else SW_CHECK_ARG(false, "gptq_gemm: no 128-row tile variant %d", t_variant);
#undef SW_GPTQ_T128
    SW_CHECK_LAUNCH();
    return SW_OK;
  }
  const int row_tiles = (M + 32 * mt - 1) / (32 * mt);
  if (tiled) {
    const int tile_cols = 32 * cp * waves;
    const int col_tiles = (N + tile_cols - 1) / tile_cols;
    int ks = pick_ksplit(col_tiles * row_tiles, K / chunk, 2, M, N, avail, 1, 256);
    if (force_ks > 0 && force_ks <= K / chunk) ks = force_ks;
    const dim3 grid((col_tiles * row_tiles * ks + 7) / 8 * 8);
    const size_t lds = (size_t)2 * 32 * mt * (chunk * 2 + 16);
#define SW_GPTQ_TILED(MT_, CP_, CHUNK_, WAVES_)                                                                 \
  do {                                                                                                          \
    auto kern = gptq_gemm_tiled_kernel<MT_, CP_, CHUNK_, WAVES_>;                                               \
    if (lds > 64 * 1024)                                                                                        \
      SW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
    hipLaunchKernelGGL(kern, grid, dim3(WAVES_ * 64), lds, st, out, ap, qweight, qzeros, sp, M, N, K,           \
                       group_size, ks, col_tiles, row_tiles, zoff, partials, counters);                         \
  } while (0)
    if (waves == 8) {
      if (mt == 1) SW_GPTQ_TILED(1, 2, 256, 8);
      else SW_GPTQ_TILED(2, 2, 256, 8);
    } else {
#ifdef SW_QUANT_EXPERIMENT
      if (mt == 2 && chunk == 256) SW_GPTQ_TILED(2, 2, 256, 4);
      else if (mt == 4 && chunk == 256) SW_GPTQ_TILED(4, 2, 256, 4);
      else if (mt == 4) SW_GPTQ_TILED(4, 2, 128, 4);
      else if (mt == 8) SW_GPTQ_TILED(8, 2, 128, 4);
      else SW_CHECK_ARG(false, "no such tiled variant");
#endif
    }
#undef SW_GPTQ_TILED
  } else {
    const int col_tiles = (N + 32 * cp - 1) / (32 * cp);
    const int ks = pick_ksplit(col_tiles * row_tiles, K / 32, 4, M, N, avail, kGemmWaves, 256);
    const dim3 grid(col_tiles * row_tiles, 1, ks);
    const size_t lds = (size_t)kGemmWaves * mt * cp * 8 * 64 * 4;
#define SW_GPTQ_KWAVES(MT_, CP_, PF_)                                                                             \
  do {                                                                                                            \
    auto kern = gptq_gemm_kwaves_kernel<MT_, CP_, PF_>;                                                           \
    if (lds > 64 * 1024)                                                                                          \
      SW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                       \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                    \
    hipLaunchKernelGGL(kern, grid, dim3(kGemmWaves * 64), lds, st, out, ap, qweight, qzeros, sp, M, N, K,         \
                       group_size, ks, row_tiles, zoff, partials, counters);                                      \
  } while (0)
    if (mt == 1 && cp == 2) SW_GPTQ_KWAVES(1, 2, 4);
    else if (mt == 1) SW_GPTQ_KWAVES(1, 4, 4);
    else if (mt == 2) SW_GPTQ_KWAVES(2, 2, 4);
    else SW_GPTQ_KWAVES(4, 2, 2);
#undef SW_GPTQ_KWAVES
  }
  SW_CHECK_LAUNCH();
  return SW_OK;
}

int sw_gptq_gemm(void* c, const void* a, const uint32_t* qweight, const uint32_t* qzeros, const void* scales,
                 const int32_t* g_idx, int M, int N, int K, int groups, int use_exllama, int bit, void* workspace,
                 int64_t workspace_bytes, void* stream) {
  return kmajor_gemm(c, a, qweight, qzeros, scales, g_idx, M, N, K, groups, use_exllama, bit, 1, workspace,
                     workspace_bytes, stream);
}

int sw_awq_to_kmajor(uint32_t* qweight_out, uint32_t* qzeros_out, const uint32_t* qweight, const uint32_t* qzeros,
                     int K, int N, int groups, void* stream) {
  SW_CHECK_ARG(K > 0 && N > 0 && groups > 0 && K % 8 == 0 && N % 8 == 0, "awq_to_kmajor: bad shape K=%d N=%d groups=%d",
               K, N, groups);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t words = (int64_t)(K / 8) * N;
  hipLaunchKernelGGL(awq_to_kmajor_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, qweight_out, qweight,
                     K / 8, N);
  SW_CHECK_LAUNCH();
  const int64_t zwords = (int64_t)groups * (N / 8);
  hipLaunchKernelGGL(awq_zeros_to_kmajor_kernel, dim3((unsigned)((zwords + 255) / 256)), dim3(256), 0, st, qzeros_out,
                     qzeros, zwords);
  SW_CHECK_LAUNCH();
  return SW_OK;
}

int sw_int4_kmajor_gemm(void* c, const void* a, const uint32_t* qweight, const uint32_t* qzeros, const void* scales,
                        int M, int N, int K, int groups, int zero_offset, void* workspace, int64_t workspace_bytes,
                        void* stream) {
  SW_CHECK_ARG(zero_offset == 0 || zero_offset == 1, "int4_kmajor_gemm: zero offset must be 0 or 1 (got %d)", zero_offset);
  return kmajor_gemm(c, a, qweight, qzeros, scales, nullptr, M, N, K, groups, 1, 4, zero_offset, workspace,
                     workspace_bytes, stream);
}

int sw_int4_kmajor_dequant(void* out, const uint32_t* qweight, const uint32_t* qzeros, const void* scales, int K, int N,
                           int groups, int zero_offset, void* stream) {
  SW_CHECK_ARG(zero_offset == 0 || zero_offset == 1, "int4_kmajor_dequant: zero offset must be 0 or 1 (got %d)",
               zero_offset);
  return kmajor_dequant(out, qweight, qzeros, scales, nullptr, K, N, groups, 1, 4, zero_offset, stream);
}

int sw_awq_dequantize(void* out, const uint32_t* qweight, const void* scales, const uint32_t* qzeros, int K, int N,
                      int group_size, void* stream) {
  SW_CHECK_ARG(K > 0 && N > 0 && N % 8 == 0 && group_size > 0 && K % group_size == 0,
               "awq_dequantize: bad shape K=%d N=%d group=%d", K, N, group_size);
  const int64_t threads = (int64_t)K * (N / 8);
  hipLaunchKernelGGL(awq_dequantize_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<uint16_t*>(out), qweight,
                     static_cast<const uint16_t*>(scales), qzeros, K, N, group_size);
  SW_CHECK_LAUNCH();
  return SW_OK;
}

int sw_awq_gemm(void* c, const void* a, const uint32_t* qweight, const void* scales, const uint32_t* qzeros, int M,
                int K, int N, int group_size, void* workspace, int64_t workspace_bytes, void* stream) {
  // the reference's argument checks (gemm_kernels.cu:671-678)
  SW_CHECK_ARG(N % 64 == 0, "OC is not multiple of cta_N = 64");
  SW_CHECK_ARG(group_size > 0 && group_size % 32 == 0, "Group size should be a multiple of 32");
  SW_CHECK_ARG(N % group_size == 0, "OC is not multiple of Group size");
  SW_CHECK_ARG(M >= 0 && K > 0 && K % group_size == 0, "awq_gemm: bad shape M=%d K=%d group=%d", M, K, group_size);
  if (M == 0) return SW_OK;
  SW_CHECK_ARG(workspace != nullptr && workspace_bytes >= kCounterBytes,
               "awq_gemm: workspace of at least %d bytes required", kCounterBytes);
  int32_t* counters = static_cast<int32_t*>(workspace);
  float* partials = reinterpret_cast<float*>(static_cast<char*>(workspace) + kCounterBytes);
  const int col_tiles = (N + 255) / 256, row_tiles = (M + 31) / 32;
  // one workgroup per compute unit (64 KB of LDS, 238 registers) and a single round of them:
  // a second, partly filled round costs more than the idle units of a first one that is not full
  const int tiles = col_tiles * row_tiles;
  const int ks = pick_ksplit(tiles, K / 16, 4, M, N, workspace_bytes, kAwqWaves, tiles * (256 / tiles > 1 ? 256 / tiles : 1));
  const size_t lds = (size_t)kAwqWaves * 8 * 4 * 64 * 4;        // one pass of the cross-wave reduction
  hipLaunchKernelGGL(awq_gemm_kernel, dim3(col_tiles * row_tiles, 1, ks), dim3(kAwqWaves * 64), lds,
                     static_cast<hipStream_t>(stream), static_cast<uint16_t*>(c), static_cast<const uint16_t*>(a),
                     qweight, static_cast<const uint16_t*>(scales), qzeros, M, N, K, group_size, ks, row_tiles,
                     partials, counters);
  SW_CHECK_LAUNCH();
  return SW_OK;
}

}  // extern "C"
