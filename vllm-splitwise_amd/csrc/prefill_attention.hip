// Causal self-attention of a prompt run (flash-style, one pass, no score matrix in
// memory) on gfx950 matrix cores.
//
// What it replaces: the prompt branch of the reference's attention backends
// (vllm/model_executor/layers/attention/backends/xformers.py:84-177, flash_attn.py:81-114),
// whose arithmetic lives in third-party xformers / flash-attn; the semantics are those
// of _ref_masked_attention (xformers.py:236-261): per sequence, softmax(scale * Q K^T +
// causal mask [+ ALiBi]) V with the probabilities rounded to the activation dtype, GQA by
// kv_head = head / (H / KVH).  Sequences are given as (start row, length) pairs, so both
// the reference's padded [B, max_len] batch and a packed batch are accepted.
//
// Tiling (wave64, v_mfma_f32_32x32x16_{bf16,f16}):
//   * a wave owns 32 query rows of one (sequence, head); 4 independent waves per workgroup.
//   * scores are produced TRANSPOSED, S^T = K . Q^T, so a lane holds one query column and
//     16 keys in registers: the softmax max/sum are lane-local plus one exchange with
//     lane^32, and the running rescale of the output needs no cross-lane traffic.
//   * the S^T accumulator, converted to 16-bit pairs, is directly the B operand of
//     O^T += V^T . P^T (32x32 accumulator rows are the next product's k index); the A
//     operand V^T is read from a row-major LDS image of the V tile with the transposing
//     ds_read_b64_tr_b16, 4 keys per read.
//   * the 4 waves of a workgroup (128 query rows) share every stage of 64 keys: K and V rows go
//     global -> registers (while the previous stage is in use) -> LDS, one barrier per stage;
//     K fragments are 16-byte LDS reads (8 contiguous head elements of one key per lane).
//   * a wave works on the stage's two score tiles together: two independent chains of Q.K products,
//     one softmax update over 32 scores per lane, then the P.V products of both with the exponentials
//     spread among them.  The vector work of a tile (~100 instructions after the measures below) is of
//     the size of its 16 matrix instructions, and a SIMD holds two waves: what one wave leaves idle, the
//     other mostly fills (counters: matrix pipe busy a third of the time, vector pipe two fifths).
//   * scores live in the log2 domain (one v_exp_f32 per probability); the output accumulator
//     is rescaled only when some lane's running maximum has grown by more than 2^8.
//   * a cached prefix (sw_context_attention) is handled before the causal part, its K and V^T
//     fragments loaded straight from the paged cache layouts.
//   * launches of less than one workgroup per CU (KSW = 2): the workgroup's waves are pairs that share 32 query rows
//     and take one score tile of every stage each, then join their (maximum, sum, output) through LDS: half as many
//     query rows per workgroup, so twice the workgroups, each wave with half the matrix work of its key loop.
#include "common.cuh"
#include <float.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace sw {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using i16x4 = __attribute__((ext_vector_type(4))) short;
using lds_i16x4_ptr = __attribute__((address_space(3))) i16x4*;

template <typename act_t> struct Mfma;
template <> struct Mfma<bf16_t> {
  using frag = bf16x8;
  __device__ static __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ __bf16 cvt(float v) { return (__bf16)v; }
};
template <> struct Mfma<f16_t> {
  using frag = f16x8;
  __device__ static __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ _Float16 cvt(float v) { return (_Float16)v; }
};

constexpr int kPfWaves = 4;
constexpr int kPfRows = 32;                          // query rows per wave
constexpr int kPfKeys = 32;                          // keys per score tile (one MFMA output block)
// score tiles per shared LDS stage (= per iteration): two while the registers allow it
__host__ __device__ constexpr int pf_tiles(int head) { return head <= 128 ? 2 : 1; }

__host__ __device__ constexpr int pf_dp(int head) { return (head + 31) / 32 * 32; }
__host__ __device__ constexpr int pf_vstride(int head) {          // bytes per V row in LDS
  return (pf_dp(head) * 2 + 255) / 256 * 256 + 64;
}
__host__ __device__ constexpr int pf_kstride(int head) {          // bytes per K row in LDS: 16 rows on 64 distinct banks
  return head * 2 + 16;
}
constexpr int kQkAhead = 4;                          // K fragments requested ahead of the product that uses them
constexpr int kPfBufs = 2;                           // LDS stages: one in use, one landing
__host__ __device__ constexpr int pf_lds_bytes(int head) {        // kPfBufs stages of a K and a V image
  return kPfBufs * pf_tiles(head) * kPfKeys * (pf_kstride(head) + pf_vstride(head));
}

// Keys and values that are already in the paged cache (a shared prefix, or the earlier chunks of
// a prompt): every new token attends to all `ctx_lens[seq]` of them before the causal part.
struct PfContext {
  const uint16_t* k_cache;        // [blocks, KVH, HEAD/8, block_size, 8]
  const uint16_t* v_cache;        // [blocks, KVH, HEAD, block_size]
  const int32_t* block_tables;    // [num_seqs, max_blocks]
  const int32_t* ctx_lens;        // [num_seqs]
  int max_blocks, block_size;
  int64_t k_block_stride, k_head_stride, v_block_stride, v_head_stride;   // elements
  // Split form (num_parts > 1): the key stages of a (query tile, sequence, head) -- cached context first, then the
  // new tokens -- are dealt out to num_parts workgroups (blockIdx.y); each leaves its unnormalised output, running
  // maximum and sum per query row here, and context_attention_reduce_kernel joins them.
  int num_parts;
  float* part_o;                  // [num_parts, total_rows, num_heads, HEAD]
  float* part_m;                  // [num_parts, total_rows, num_heads]   (log2 domain)
  float* part_l;                  // [num_parts, total_rows, num_heads]
  int64_t total_rows;
};

template <typename act_t, int HEAD, bool CTX, int WAVES, int KSW>
__global__ void __launch_bounds__(WAVES * 64, (HEAD <= 128 && WAVES == 4) ? 2 : 1)
prefill_attention_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ q,
                         const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                         const int32_t* __restrict__ seq_starts, const int32_t* __restrict__ seq_lens,
                         int num_heads, int num_kv_heads, float scale,
                         const float* __restrict__ alibi_slopes, int64_t q_stride, int64_t k_stride,
                         int64_t v_stride, int64_t out_stride, int sliding_window, int num_seqs, int max_seq_len,
                         PfContext cx) {
  using M = Mfma<act_t>;
  using frag = typename M::frag;
  constexpr int NT = pf_tiles(HEAD);                 // score tiles per stage
  constexpr int NTW = NT / KSW;                      // ... of which a wave takes this many
  constexpr int RG = WAVES / KSW;                    // groups of 32 query rows per workgroup
  static_assert(NT % KSW == 0 && WAVES % KSW == 0 && (KSW == 1 || KSW == 2), "key split of a stage's score tiles");
  constexpr int kPfStage = NT * kPfKeys;             // keys per stage
  constexpr int KS = HEAD / 16;                      // k-steps of the Q.K product
  constexpr int DP = pf_dp(HEAD);
  constexpr int DB = DP / 32;                        // 32-row blocks of O^T
  constexpr int VSTRIDE = pf_vstride(HEAD);
  constexpr int CPR = HEAD / 8;                      // 16-byte chunks per V row
  constexpr int VLOADS = kPfKeys * CPR / 64;         // V chunks per lane per tile
  static_assert(HEAD % 16 == 0 && (kPfKeys * CPR) % 64 == 0, "unsupported head size");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KSTRIDE = pf_kstride(HEAD);
  constexpr int CHUNKS = kPfStage * CPR;             // 16-byte pieces of one stage of K (and of V)
  constexpr int LOADS = (CHUNKS + WAVES * 64 - 1) / (WAVES * 64);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* k_img = smem;                                // [kPfBufs][kPfStage][KSTRIDE]
  char* v_img = smem + kPfBufs * kPfStage * KSTRIDE; // [kPfBufs][kPfStage][VSTRIDE]

  // One workgroup per (query tile, sequence, head), numbered tile-major with the LATEST tiles first: under the
  // causal mask they have the longest key loops, so the launch's last round is made of the short ones.  Within
  // a tile index the heads that share a KV head are 8 workgroup numbers apart (workgroups go round-robin over
  // the 8 XCDs), so that one L2 serves the whole group's reads of its keys and values.
  const int per_tile = num_heads * num_seqs;
  const int tile_rank = blockIdx.x / per_tile;
  const int rem = blockIdx.x - tile_rank * per_tile;
  const int seq = rem / num_heads;
  const int hj = rem - seq * num_heads;
  const int head = (hj % num_kv_heads) * (num_heads / num_kv_heads) + hj / num_kv_heads;
  const int seq_len = seq_lens[seq];
  // Query tiles are aligned to the END of the sequence: tile e (from the end) covers the rows
  // [seq_len - (e+1) R, seq_len - e R), R rows per workgroup, and the sequence's FIRST tile takes what is left.
  // The remainder rows then sit in the tile with the shortest key loop, not in one that walks all the keys
  // with most of its waves idle, and every later tile is full.
  // Workgroup rank r (0 = first to start) holds, of every sequence, the tile whose end is r tiles before the end of
  // the LONGEST sequence of the launch: a short sequence's tiles start when the long ones' tiles of the same key
  // count do, so that ragged launches too begin with their heaviest tiles.
  const int tile_e = tile_rank - (max_seq_len - seq_len) / (RG * kPfRows);
  if (tile_e < 0) return;                            // workgroup-uniform
  const int q_end = seq_len - tile_e * RG * kPfRows;
  if (q_end <= 0) return;                            // workgroup-uniform
  const int q_base = max(0, q_end - RG * kPfRows);
  const int kw = wave % KSW;                         // which of a stage's tiles this wave takes (tiles kw * NTW ..)
  const int q0 = q_base + (wave / KSW) * kPfRows;
  const bool active = q0 < q_end;                    // wave-uniform; idle waves still load tiles and meet barriers
  const int64_t row0 = seq_starts[seq];
  const int kv_head = head / (num_heads / num_kv_heads);
  const float slope = alibi_slopes ? alibi_slopes[head] : 0.f;

  const int col = lane & 31;                         // query column of S^T / O^T held by this lane
  const int half = lane >> 5;
  const int q_idx = q0 + col;                        // position inside the sequence

  // ---- Q^T fragments (B operand, natural k order): Q[q][16*ks + 8*half + j]
  frag qf[KS];
  {
    const int qr = min(q_idx, q_end - 1);
    const uint16_t* qp = q + (row0 + qr) * q_stride + (int64_t)head * HEAD + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const frag*>(qp + 16 * ks);
  }

  // zero the V images once: pad columns HEAD..DP-1 must read as 0 (never written again)
  if constexpr (DP != HEAD) {
    for (int i = tid; i < kPfBufs * kPfStage * VSTRIDE / 16; i += WAVES * 64)
      reinterpret_cast<uint4*>(v_img)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
  }

  f32x16 o[DB];
#pragma unroll
  for (int b = 0; b < DB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
  float m_run = -INFINITY, l_part = 0.f;

  // transposing-read addressing (see header): lane 4q+p of a 16-lane group supplies row q,
  // columns 4p..4p+3 of a 4 x 16 block; group g covers O^T rows 16*(g&1).. of key half g>>1
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = lane >> 4;
  const int tr_off = (4 * (tr_g >> 1) + tr_q) * VSTRIDE + (16 * (tr_g & 1) + 4 * tr_p) * 2;
  const int k_off = col * KSTRIDE + half * 16;       // this lane's K fragment inside a score tile

  // Scores are kept in the log2 domain (scale and ALiBi slope carry a factor log2 e), so a
  // probability is one v_exp_f32.  Online-softmax update shared by both phases: `s` holds the
  // masked, scaled scores of 32 keys.
  const float scale2 = scale * 1.4426950408889634f, slope2 = slope * 1.4426950408889634f;
  // Online-softmax update over NT score tiles (32 keys each), in parts.  tile_max: the tiles' maximum
  // joined with the running one.  wants_rescale / rescale: the output accumulator and the running sum
  // are referred to a new maximum only when some lane's has grown by more than 2^8 since they last
  // were; until then the probabilities are taken against that older maximum and may exceed 1 by up to
  // that factor, which the fp32 sums and the 16-bit fragments hold without loss (the common factor
  // cancels in the final division).  softmax_exp: the probabilities against `m_ref()`, as B operands
  // of the P.V products (two 16-key fragments per tile), and their sum.
  // the value held by lane ^ 32 (one v_permlane32_swap; a shuffle through LDS would stall the wave for the
  // length of an LDS round trip in every stage)
  auto other_half = [](float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
  };
  // S^T = K . Q^T of the NT score tiles of a stage whose K image starts at `kb` (this lane's fragment offset
  // included).  The tiles' chains of matrix products alternate, and the order of LDS reads and matrix
  // instructions is pinned: kQkAhead fragments are requested before the first product and one more after each
  // product, so that a product never waits for a read issued just before it.
  auto qk_stage = [&](const char* kb, f32x16* sv) {
    constexpr int NF = NTW * KS;
    constexpr int AHEAD = NF < kQkAhead ? NF : kQkAhead;
    frag kf[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
      kf[i] = *reinterpret_cast<const frag*>(kb + (i % NTW) * kPfKeys * KSTRIDE + 32 * (i / NTW));
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[u][r] = 0.f;
#pragma unroll
    for (int i = 0; i < NF; ++i) sv[i % NTW] = M::run(kf[i], qf[i / NTW], sv[i % NTW]);
    __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
    for (int i = 0; i < NF - AHEAD; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 0);
  };
  auto tile_max = [&](auto NTC, const f32x16* s, const float* mult) {
    constexpr int NTS = decltype(NTC)::value;
    float m_tile = -INFINITY;
#pragma unroll
    for (int u = 0; u < NTS; ++u) {
      float m_u = s[u][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m_u = fmaxf(m_u, s[u][r]);
      m_tile = fmaxf(m_tile, m_u * mult[u]);           // mult > 0: the maximum commutes with it
    }
    m_tile = fmaxf(m_tile, other_half(m_tile));
    return fmaxf(m_run, m_tile);
  };
  auto wants_rescale = [&](float m_new) { return __builtin_amdgcn_ballot_w64(m_new > m_run + 8.0f) != 0; };
  auto rescale = [&](float m_new) {
    const float alpha = __builtin_amdgcn_exp2f(m_run - ((m_new == -INFINITY) ? 0.f : m_new));
    l_part *= alpha;
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
    m_run = m_new;
  };
  // a query row beyond the sequence (or a fully masked window tile) keeps m at -inf: use 0
  auto m_ref = [&]() { return (m_run == -INFINITY) ? 0.f : m_run; };
  // Tile u's scores are s[u] * mult[u]: a tile with nothing to mask and no bias keeps the raw products and
  // mult = the (log2-domain) softmax scale, which then rides in the exponent's multiply-add; a tile that was
  // scaled, biased or masked element by element comes with mult = 1.
  auto softmax_exp = [&](auto NTC, const f32x16* s, const float* mult, float m_safe, frag* pf) {
    constexpr int NTS = decltype(NTC)::value;
    float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
    for (int u = 0; u < NTS; ++u)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(s[u][r], mult[u], -m_safe));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(s[u][r + 1], mult[u], -m_safe));
        psum0 += p0;
        psum1 += p1;
        pf[2 * u + (r >> 3)][r & 7] = M::cvt(p0);      // rounded to the activation dtype
        pf[2 * u + (r >> 3)][(r & 7) + 1] = M::cvt(p1);
      }
    l_part += psum0 + psum1;
  };
  auto softmax_step = [&](auto NTC, const f32x16* s, const float* mult, frag* pf) {
    const float m_new = tile_max(NTC, s, mult);
    if (wants_rescale(m_new)) rescale(m_new);
    softmax_exp(NTC, s, mult, m_ref(), pf);
  };
  using One = std::integral_constant<int, 1>;

  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  // the new tokens' key stages of this workgroup (phase 2), and the part of all stages that this workgroup takes
  const int wg_end = q_end;
  int wg_begin = 0;
  if (sliding_window > 0) wg_begin = max(0, (q_base - sliding_window + 1) / kPfStage * kPfStage);
  const int stages_all = (wg_end - wg_begin + kPfStage - 1) / kPfStage;
  const int num_parts = CTX ? max(cx.num_parts, 1) : 1;
  const int part = CTX ? (int)blockIdx.y : 0;
  int t_lo = 0, t_hi = 0x7fffffff, s_lo = 0, s_hi = stages_all;
  if (CTX && num_parts > 1) {                          // workgroup-uniform
    const int ctx = cx.ctx_lens[seq];
    int c0 = 0;
    if (sliding_window > 0) c0 = max(0, (ctx + q_base - sliding_window + 1) / kPfStage * kPfStage);
    const int cst = ctx > c0 ? (ctx - c0 + kPfStage - 1) / kPfStage : 0;
    const int total = cst + stages_all;
    const int g_lo = (int)((int64_t)total * part / num_parts), g_hi = (int)((int64_t)total * (part + 1) / num_parts);
    t_lo = min(g_lo, cst);
    t_hi = min(g_hi, cst);
    s_lo = max(g_lo - cst, 0);
    s_hi = max(g_hi - cst, 0);
  }
  // The shared form of phase 1 needs 8 consecutive keys of a head element in one block and no pad rows in O^T
  const bool ctx_shared = CTX && DP == HEAD && cx.block_size % 8 == 0;
  if (CTX && ctx_shared) {
    // ---- phase 1, shared: the cached context goes through LDS in stages of kPfStage keys, like the new tokens in
    // phase 2, so the workgroup's waves read every cached key and value once between them instead of once each.
    // K pieces (16 bytes = 8 head elements of a key) land in the row-major K image of phase 2.  The value cache is
    // already V^T ([head element][key]), which is what the P.V product wants as its A operand: its 16-byte pieces
    // (8 keys of a head element) build a [head element][key] image in the V buffers, read back with plain 8-byte
    // reads -- no transposing read here.
    constexpr int VTS = kPfStage * 2 + 8;              // bytes per V^T row: 32 rows on distinct bank pairs
    static_assert(HEAD * VTS <= kPfStage * VSTRIDE, "the V^T image must fit a V buffer");
    const int ctx = cx.ctx_lens[seq];
    const int32_t* table = cx.block_tables + (int64_t)seq * cx.max_blocks;
    const int bs = cx.block_size;
    const int q_abs = ctx + q_idx;
    int c0 = 0;
    if (sliding_window > 0) c0 = max(0, (ctx + q_base - sliding_window + 1) / kPfStage * kPfStage);
    const int cstages_all = ctx > c0 ? (ctx - c0 + kPfStage - 1) / kPfStage : 0;   // workgroup-uniform
    const int cfirst = c0 + t_lo * kPfStage;                                       // this workgroup's share of them
    const int cstages = min(t_hi, cstages_all) - t_lo;
    u32x4 ckreg[LOADS], cvreg[LOADS];
    // The block numbers of a stage's pieces are read one stage before the pieces themselves (tfetch before cfetch): a
    // piece's address depends on its block-table entry, and a wave that has to wait for that entry before it can even
    // issue the loads stands still for a memory round trip in every stage.
    int tkb[LOADS], tvb[LOADS];
    auto tfetch = [&](int k0) {
      static_for<LOADS>([&](auto IT) {
        constexpr int it = decltype(IT)::value;
        const int c = it * WAVES * 64 + tid;
        if (CHUNKS % (WAVES * 64) == 0 || c < CHUNKS) {
          tkb[it] = table[min(k0 + c / CPR, ctx - 1) / bs];
          tvb[it] = table[(min(k0 + 8 * (c % (kPfStage / 8)), ctx - 1) & ~7) / bs];
        }
      });
    };
    auto cfetch = [&](int k0) {
      static_for<LOADS>([&](auto IT) {
        constexpr int it = decltype(IT)::value;
        const int c = it * WAVES * 64 + tid;
        if (CHUNKS % (WAVES * 64) == 0 || c < CHUNKS) {
          const int kk = min(k0 + c / CPR, ctx - 1), ch = c % CPR;       // keys past the context: any valid address
          const uint16_t* kp = cx.k_cache + (int64_t)tkb[it] * cx.k_block_stride +
                               (int64_t)kv_head * cx.k_head_stride + (int64_t)ch * bs * 8 + (kk % bs) * 8;
          ckreg[it] = *reinterpret_cast<const u32x4*>(kp);
          const int d = c / (kPfStage / 8), key0 = k0 + 8 * (c % (kPfStage / 8));
          const int kc = min(key0, ctx - 1) & ~7;
          const uint16_t* vp = cx.v_cache + (int64_t)tvb[it] * cx.v_block_stride +
                               (int64_t)kv_head * cx.v_head_stride + (int64_t)d * bs + (kc % bs);
          u32x4 w = *reinterpret_cast<const u32x4*>(vp);
          const int valid = ctx - key0;                                  // never-written slots may hold NaN patterns
          if (valid < 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (2 * i + 0 >= valid) w[i] &= 0xffff0000u;
              if (2 * i + 1 >= valid) w[i] &= 0x0000ffffu;
            }
          }
          cvreg[it] = w;
        }
      });
    };
    auto ccommit = [&](int buf) {
      static_for<LOADS>([&](auto IT) {
        constexpr int it = decltype(IT)::value;
        const int c = it * WAVES * 64 + tid;
        if (CHUNKS % (WAVES * 64) == 0 || c < CHUNKS) {
          const int key = c / CPR, ch = c - key * CPR;
          *reinterpret_cast<u32x4*>(k_img + (buf * kPfStage + key) * KSTRIDE + ch * 16) = ckreg[it];
          const int d = c / (kPfStage / 8), g8 = c % (kPfStage / 8);
          *reinterpret_cast<u32x4*>(v_img + buf * kPfStage * VSTRIDE + d * VTS + g8 * 16) = cvreg[it];
        }
      });
    };
    if (cstages > 0) {
      tfetch(cfirst);
      cfetch(cfirst);
      if (cstages > 1) tfetch(cfirst + kPfStage);
      ccommit(0);
      if (cstages > 1) cfetch(cfirst + kPfStage);
      if (cstages > 2) tfetch(cfirst + 2 * kPfStage);
      __syncthreads();
    }
    int cbuf = 0;
    for (int t = 0; t < cstages; ++t) {                  // every wave meets every barrier
      const int k0 = cfirst + t * kPfStage;
      if (t + 1 < cstages) ccommit(cbuf ^ 1);
      if (t + 2 < cstages) cfetch(k0 + 2 * kPfStage);
      if (t + 3 < cstages) tfetch(k0 + 3 * kPfStage);
      // (wave-uniform) this wave's window may begin after the stage
      const bool wanted = active && (sliding_window <= 0 || k0 + kPfStage > ctx + q0 - sliding_window + 1);
      if (wanted) {
        const char* kb = k_img + (cbuf * kPfStage + kw * NTW * kPfKeys) * KSTRIDE + col * KSTRIDE + half * 16;
        const char* vt = v_img + cbuf * kPfStage * VSTRIDE + col * VTS + 8 * half + kw * NTW * kPfKeys * 2;
        f32x16 sv[NTW];
        qk_stage(kb, sv);
        float mult[NTW];
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
          const int kt = k0 + (kw * NTW + u) * kPfKeys;
          mult[u] = scale2;
          if (!(kt + kPfKeys <= ctx && slope == 0.f && sliding_window <= 0)) {   // wave-uniform: something to mask
            mult[u] = 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = kt + (r & 3) + 8 * (r >> 2) + 4 * half;
              float x = sv[u][r] * scale2;
              if (slope != 0.f) x += slope2 * (float)(key - q_abs);
              bool ok = key < ctx;
              if (sliding_window > 0) ok = ok && (q_abs - key < sliding_window);
              sv[u][r] = ok ? x : -INFINITY;
            }
          }
        }
        frag pt[2 * NTW];
        softmax_step(std::integral_constant<int, NTW>{}, sv, mult, pt);
        // O^T += V^T . P^T; A operand element j <-> key 16*st + 8*(j>>2) + 4*half + (j&3) of the tile
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
          for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int b = 0; b < DB; ++b) {
              const char* base = vt + b * 32 * VTS + (32 * u + 16 * st) * 2;
              union { uint2 p[2]; frag f; } vf;
              vf.p[0] = *reinterpret_cast<const uint2*>(base);
              vf.p[1] = *reinterpret_cast<const uint2*>(base + 16);
              o[b] = M::run(vf.f, pt[2 * u + st], o[b]);
            }
      }
      __syncthreads();
      cbuf ^= 1;
    }
  } else if (CTX && active) {
    // ---- phase 1, direct (odd block sizes and head sizes): the cached context.  Both operands come straight from the cache layouts: 8
    // contiguous head elements of a key are one 16-byte group of the key cache, and 4 contiguous
    // keys of one head element are 8 bytes of the value cache (= a piece of V^T).
    const int ctx = cx.ctx_lens[seq];
    const int32_t* table = cx.block_tables + (int64_t)seq * cx.max_blocks;
    const int bs = cx.block_size;
    const int q_abs = ctx + q_idx;
    int c_begin = 0;
    if (sliding_window > 0) c_begin = max(0, (ctx + q0 - sliding_window + 1) / kPfKeys * kPfKeys);
    // (waves in pairs take alternate tiles)
    for (int k0 = c_begin + kw * kPfKeys; k0 < ctx; k0 += KSW * kPfKeys) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      {
        const int kk = min(k0 + col, ctx - 1);
        const uint16_t* kp = cx.k_cache + (int64_t)table[kk / bs] * cx.k_block_stride +
                             (int64_t)kv_head * cx.k_head_stride + (kk % bs) * 8 + (int64_t)half * bs * 8;
        frag kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const frag*>(kp + (int64_t)(2 * ks) * bs * 8);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s = M::run(kf[ks], qf[ks], s);
      }
      // V^T pieces: O^T row d = 32 b + col, keys k0 + 16 st + 8 g + 4 half .. + 3
      uint2 vraw[2][DB][2];
      const bool tail = k0 + kPfKeys > ctx;
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int key0 = k0 + 16 * st + 8 * g + 4 * half;
          const int kc = min(key0, ctx - 1);
          const uint16_t* vp = cx.v_cache + (int64_t)table[kc / bs] * cx.v_block_stride +
                               (int64_t)kv_head * cx.v_head_stride + (kc % bs);
#pragma unroll
          for (int b = 0; b < DB; ++b) {
            const int d = 32 * b + col;
            uint2 w = make_uint2(0, 0);
            if (d < HEAD && key0 < ctx) w = *reinterpret_cast<const uint2*>(vp + (int64_t)d * bs);
            if (tail) {                                // never-written slots may hold NaN bit patterns
              if (key0 + 0 >= ctx) w.x &= 0xffff0000u;
              if (key0 + 1 >= ctx) w.x &= 0x0000ffffu;
              if (key0 + 2 >= ctx) w.y &= 0xffff0000u;
              if (key0 + 3 >= ctx) w.y &= 0x0000ffffu;
            }
            vraw[st][b][g] = w;
          }
        }
      if (!tail && slope == 0.f && sliding_window <= 0) {          // wave-uniform: nothing to mask
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] *= scale2;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          float x = s[r] * scale2;
          if (slope != 0.f) x += slope2 * (float)(key - q_abs);
          bool ok = key < ctx;
          if (sliding_window > 0) ok = ok && (q_abs - key < sliding_window);
          s[r] = ok ? x : -INFINITY;
        }
      }
      frag pf[2];
      const float one = 1.f;
      softmax_step(One{}, &s, &one, pf);
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int b = 0; b < DB; ++b) {
          union { uint2 u[2]; frag f; } vf;
          vf.u[0] = vraw[st][b][0];
          vf.u[1] = vraw[st][b][1];
          o[b] = M::run(vf.f, pf[st], o[b]);
        }
    }
  }

  // ---- phase 2: the new tokens, causal.  The workgroup's 4 waves (128 query rows) share every
  // K / V stage of 64 keys through LDS: global -> registers while the previous stage is being
  // used, registers -> LDS, one barrier per stage.
  const int k_end = min(q0 + kPfRows, q_end);        // this wave's keys: < k_end
  int k_begin = 0;
  if (sliding_window > 0) k_begin = max(0, (q0 - sliding_window + 1) / kPfKeys * kPfKeys);

  // (native vectors: arrays of HIP's uint4 struct, assigned under a condition, are kept in scratch
  //  memory by the compiler)
  u32x4 kreg[LOADS], vreg[LOADS];
  // K and V rows of this (sequence, kv head) through buffer loads: a lane's byte offset inside a stage is
  // fixed, a stage adds a uniform amount, and rows past the end of the sequence are out of the
  // descriptor's range and read as zero (no clamping, no 64-bit address arithmetic in the loop)
  const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(k + row0 * k_stride + (int64_t)kv_head * HEAD), 0,
      (int)(uint32_t)(((int64_t)(seq_len - 1) * k_stride + HEAD) * 2), 0x00020000);
  const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(v + row0 * v_stride + (int64_t)kv_head * HEAD), 0,
      (int)(uint32_t)(((int64_t)(seq_len - 1) * v_stride + HEAD) * 2), 0x00020000);
  uint32_t koff[LOADS], voff[LOADS];
#pragma unroll
  for (int it = 0; it < LOADS; ++it) {
    const int c = it * WAVES * 64 + tid;
    const int key = c / CPR, ch = c - key * CPR;
    koff[it] = (uint32_t)(key * k_stride + ch * 8) * 2u;
    voff[it] = (uint32_t)(key * v_stride + ch * 8) * 2u;
  }
  // (compile-time slot indices and no early exit from the loop body: the staging registers must
  // not end up addressed through memory)
  auto fetch = [&](int t0) {
    const uint32_t k_stage = (uint32_t)t0 * (uint32_t)k_stride * 2u, v_stage = (uint32_t)t0 * (uint32_t)v_stride * 2u;
    static_for<LOADS>([&](auto IT) {
      constexpr int it = decltype(IT)::value;
      if (CHUNKS % (WAVES * 64) == 0 || it * WAVES * 64 + tid < CHUNKS) {
        kreg[it] = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, (int)(koff[it] + k_stage), 0, 0);
        vreg[it] = __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, (int)(voff[it] + v_stage), 0, 0);
      }
    });
  };
  auto commit = [&](int buf) {
    static_for<LOADS>([&](auto IT) {
      constexpr int it = decltype(IT)::value;
      const int c = it * WAVES * 64 + tid;
      if (CHUNKS % (WAVES * 64) == 0 || c < CHUNKS) {
        const int key = c / CPR, ch = c - key * CPR;
        *reinterpret_cast<u32x4*>(k_img + (buf * kPfStage + key) * KSTRIDE + ch * 16) = kreg[it];
        *reinterpret_cast<u32x4*>(v_img + (buf * kPfStage + key) * VSTRIDE + ch * 16) = vreg[it];
      }
    });
  };

  // Stages of NT score tiles (64 keys up to head size 128), two LDS buffers:
  //   iteration t:  stage t+1 registers -> LDS (the buffer that iteration t-1 read, released by its
  //                 barrier) | stage t+2 global -> registers | both score tiles of stage t: S = K . Q^T,
  //                 mask, one softmax update, O += V^T . P^T | barrier
  const int stages = s_hi - s_lo;                      // all of them unless the stages are split over workgroups
  const int wg_first = wg_begin + s_lo * kPfStage;
  auto needs = [&](int k0) { return active && k0 < k_end && k0 + kPfKeys > k_begin; };

  if (stages > 0) {
    fetch(wg_first);
    commit(0);
    if (stages > 1) fetch(wg_first + kPfStage);
    __syncthreads();
  }
  int buf = 0;
  for (int t = 0; t < stages; ++t) {                   // stages is workgroup-uniform: every wave meets every barrier
    const int k0 = wg_first + t * kPfStage;
    if (t + 1 < stages) commit(buf ^ 1);
    if (t + 2 < stages) fetch(k0 + 2 * kPfStage);
    bool any = false;                                              // wave-uniform
#pragma unroll
    for (int u = 0; u < NTW; ++u) any = any || needs(k0 + (kw * NTW + u) * kPfKeys);
    if (any) {
      // (a tile this wave does not need -- above the diagonal, outside the window -- is computed and
      // masked: keeping it out would make the fragment registers conditionally assigned, which the
      // compiler answers with scratch memory)
      const char* kb = k_img + (buf * kPfStage + kw * NTW * kPfKeys) * KSTRIDE + k_off;
      const char* vb = v_img + (buf * kPfStage + kw * NTW * kPfKeys) * VSTRIDE + tr_off;
      // scale and mask (this lane: query q_idx, keys kt + (r&3) + 8*(r>>2) + 4*half)
      // (returns the tile's multiplier, see softmax_exp)
      auto mask_scale = [&](f32x16& sc, int kt) {
        const bool wanted = needs(kt);
        const bool need_mask = (kt + kPfKeys - 1 > q0) || (sliding_window > 0);
        if (wanted && !need_mask && slope == 0.f) return scale2;     // wave-uniform: a tile below the diagonal
        if (slope == 0.f && sliding_window <= 0) {                   // wave-uniform: only the causal mask
          const int lim = wanted ? q_idx - kt - 4 * half : -1;       // element r is visible iff (r&3) + 8*(r>>2) <= lim
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[r] = ((r & 3) + 8 * (r >> 2) <= lim) ? sc[r] : -INFINITY;
          return scale2;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt + (r & 3) + 8 * (r >> 2) + 4 * half;
          float x = sc[r] * scale2;
          if (slope != 0.f) x += slope2 * (float)(key - q_idx);
          bool ok = wanted && key <= q_idx;
          if (sliding_window > 0) ok = ok && (q_idx - key < sliding_window);
          sc[r] = ok ? x : -INFINITY;
        }
        return 1.f;
      };
      // O^T += V^T . P^T of tile u; A operand element j <-> key 16*st + 8*(j>>2) + 4*half + (j&3)
      auto pv_tile = [&](int u, const frag* pt) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
          for (int b = 0; b < DB; ++b) {
            const char* base = vb + (32 * u + 16 * st) * VSTRIDE + b * 64;
            const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4_ptr)(base));
            const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4_ptr)(base + 8 * VSTRIDE));
            union { struct { i16x4 a, b; } p; frag f; } vf;
            vf.p.a = lo;
            vf.p.b = hi;
            o[b] = M::run(vf.f, pt[st], o[b]);
          }
        }
      };
      // Both tiles' scores first (two independent chains of matrix products), ONE softmax update over
      // the 32 scores per lane (one maximum exchange, one rescale decision), then both tiles' P.V
      // products, among which the compiler spreads the exponentials.  (Tried and dropped: exp(a)
      // beside S_b and exp(b) beside P_a.V with an update per tile -- the rescale between the tiles
      // must then see tile a's products, and the two code arms that takes cost more register copies
      // than the overlap saves: the two waves of a SIMD already fill each other's gaps.)
      f32x16 sv[NTW];
      qk_stage(kb, sv);
      float mult[NTW];
#pragma unroll
      for (int u = 0; u < NTW; ++u) mult[u] = mask_scale(sv[u], k0 + (kw * NTW + u) * kPfKeys);
      frag pt[2 * NTW];
      softmax_step(std::integral_constant<int, NTW>{}, sv, mult, pt);
#pragma unroll
      for (int u = 0; u < NTW; ++u) pv_tile(u, pt + 2 * u);
    }
    __syncthreads();
    buf ^= 1;
  }
  if constexpr (KSW == 2) {
    // ---- the two waves of a row group join what they saw of the keys: the odd wave hands its running maximum, sum
    // and output to the even one through LDS (the K / V images are free: every loop above ends with a barrier)
    constexpr int XW = DB * 16 + 2;                    // floats per lane
    static_assert(RG * XW * 64 * 4 <= pf_lds_bytes(HEAD), "the exchange must fit the stage buffers");
    float* xch = reinterpret_cast<float*>(smem) + (wave / KSW) * XW * 64 + lane;
    __syncthreads();
    if (kw == 1) {
#pragma unroll
      for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(b * 16 + r) * 64] = o[b][r];
      xch[DB * 16 * 64] = m_run;
      xch[(DB * 16 + 1) * 64] = l_part;
    }
    __syncthreads();
    if (kw == 1) return;
    const float m_other = xch[DB * 16 * 64], l_other = xch[(DB * 16 + 1) * 64];
    const float m_new = fmaxf(m_run, m_other);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float a_own = __builtin_amdgcn_exp2f(m_run - m_safe), a_other = __builtin_amdgcn_exp2f(m_other - m_safe);
    l_part = l_part * a_own + l_other * a_other;
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[b][r] = o[b][r] * a_own + xch[(b * 16 + r) * 64] * a_other;
    m_run = m_new;
  }
  if (!active) return;

  // ---- normalise and store: this lane holds query q_idx, O^T rows d = 32*b + (r&3) + 8*(r>>2) + 4*half
  const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
  if (CTX && num_parts > 1) {
    // split form: this workgroup saw only some of the keys; leave (O, m, l) for the reduce kernel
    if (q_idx < q_end) {
      const int64_t idx = ((int64_t)part * cx.total_rows + row0 + q_idx) * num_heads + head;
      float* po = cx.part_o + idx * HEAD;
#pragma unroll
      for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int d = 32 * b + 8 * r4 + 4 * half;
          if (d < HEAD)
            *reinterpret_cast<float4*>(po + d) =
                make_float4(o[b][4 * r4], o[b][4 * r4 + 1], o[b][4 * r4 + 2], o[b][4 * r4 + 3]);
        }
      if (half == 0) {
        cx.part_m[idx] = m_run;
        cx.part_l[idx] = l_tot;
      }
    }
    return;
  }
  const float inv = 1.f / l_tot;
  if (q_idx < q_end) {
    uint16_t* op = out + (row0 + q_idx) * out_stride + (int64_t)head * HEAD;
#pragma unroll
    for (int b = 0; b < DB; ++b) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d = 32 * b + 8 * r4 + 4 * half;
        if (d < HEAD) {
          // (the hardware's conversions, as for the probabilities: round to nearest even in one instruction per pair
          // where the generic helper spends seven on each of a lane's 64 values -- a fifth of a short prompt's
          // vector instructions)
          uint16_t w[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = __builtin_bit_cast(uint16_t, M::cvt(o[b][4 * r4 + j] * inv));
          *reinterpret_cast<uint2*>(op + d) =
              make_uint2((uint32_t)w[0] | ((uint32_t)w[1] << 16), (uint32_t)w[2] | ((uint32_t)w[3] << 16));
        }
      }
    }
  }
}

// Joins the parts of the split form: out = sum_p O_p 2^(m_p - m) / sum_p l_p 2^(m_p - m), m = max_p m_p.
// One thread per (query row, head, 4 head elements); all of a thread's loads are issued before the first is used
// (the parts' maxima, sums and output pieces do not depend on each other), so the kernel runs at memory speed
// rather than at one round trip per part.
constexpr int kCtxReduceMaxParts = 8;                // parts joined per pass of the unrolled loads

template <typename act_t>
__global__ void __launch_bounds__(256)
context_attention_reduce_kernel(uint16_t* __restrict__ out, const float* __restrict__ part_o,
                                const float* __restrict__ part_m, const float* __restrict__ part_l,
                                const int32_t* __restrict__ seq_starts, const int32_t* __restrict__ new_lens,
                                int num_heads, int head_size, int num_parts, int64_t total_rows, int64_t out_stride) {
  const int seq = blockIdx.z;
  if ((int)blockIdx.x >= new_lens[seq]) return;
  const int64_t row = (int64_t)seq_starts[seq] + blockIdx.x;
  const int per_head = head_size / 4;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= num_heads * per_head) return;
  const int head = c / per_head, d = (c - head * per_head) * 4;
  float m = -INFINITY, l = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p0 = 0; p0 < num_parts; p0 += kCtxReduceMaxParts) {
    float mp[kCtxReduceMaxParts], lp[kCtxReduceMaxParts];
    float4 op[kCtxReduceMaxParts];
#pragma unroll
    for (int i = 0; i < kCtxReduceMaxParts; ++i) {
      const int p = min(p0 + i, num_parts - 1);          // a repeated last part is given weight 0 below
      const int64_t idx = ((int64_t)p * total_rows + row) * num_heads + head;
      mp[i] = part_m[idx];
      lp[i] = part_l[idx];
      op[i] = *reinterpret_cast<const float4*>(part_o + idx * head_size + d);
    }
    float m_new = m;
#pragma unroll
    for (int i = 0; i < kCtxReduceMaxParts; ++i)
      if (p0 + i < num_parts) m_new = fmaxf(m_new, mp[i]);
    if (m_new == -INFINITY) continue;                    // none of these parts saw a key of the row
    const float rescale = __builtin_amdgcn_exp2f(m - m_new);       // 0 while m is still -inf
    l *= rescale;
    acc.x *= rescale; acc.y *= rescale; acc.z *= rescale; acc.w *= rescale;
    m = m_new;
#pragma unroll
    for (int i = 0; i < kCtxReduceMaxParts; ++i) {
      const bool live = p0 + i < num_parts && mp[i] != -INFINITY;
      const float w = live ? __builtin_amdgcn_exp2f(mp[i] - m) : 0.f;
      l = fmaf(live ? lp[i] : 0.f, w, l);
      acc.x = fmaf(live ? op[i].x : 0.f, w, acc.x);
      acc.y = fmaf(live ? op[i].y : 0.f, w, acc.y);
      acc.z = fmaf(live ? op[i].z : 0.f, w, acc.z);
      acc.w = fmaf(live ? op[i].w : 0.f, w, acc.w);
    }
  }
  const float inv = 1.f / l;
  const uint32_t lo = (uint32_t)Conv<act_t>::from_float(acc.x * inv) | ((uint32_t)Conv<act_t>::from_float(acc.y * inv) << 16);
  const uint32_t hi = (uint32_t)Conv<act_t>::from_float(acc.z * inv) | ((uint32_t)Conv<act_t>::from_float(acc.w * inv) << 16);
  *reinterpret_cast<uint2*>(out + row * out_stride + (int64_t)head * head_size + d) = make_uint2(lo, hi);
}

template <typename act_t, int HEAD, bool CTX, int WAVES, int KSW>
int launch_prefill_w(void* out, const void* q, const void* k, const void* v, const int32_t* seq_starts,
                     const int32_t* seq_lens, int num_seqs, int max_seq_len, int num_heads, int num_kv_heads,
                     float scale, const float* alibi, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                     int64_t out_stride, int sliding_window, const PfContext& cx, hipStream_t stream) {
  const int rows_per_wg = WAVES / KSW * kPfRows;
  const int q_tiles = (max_seq_len + rows_per_wg - 1) / rows_per_wg;
  SW_CHECK_ARG((int64_t)q_tiles * num_heads * num_seqs < (int64_t)1 << 31, "prefill attention: %d x %d x %d workgroups",
               q_tiles, num_heads, num_seqs);
  // the split form needs the shared form of the context phase (see the kernel)
  const int parts = (CTX && cx.num_parts > 1 && pf_dp(HEAD) == HEAD && cx.block_size % 8 == 0) ? cx.num_parts : 1;
  SW_CHECK_ARG(parts == std::max(cx.num_parts, 1), "context_attention: head size %d / block size %d cannot be split",
               HEAD, cx.block_size);
  dim3 grid((unsigned)(q_tiles * num_heads * num_seqs), (unsigned)parts);
  const size_t smem = (size_t)pf_lds_bytes(HEAD);
  auto kern = prefill_attention_kernel<act_t, HEAD, CTX, WAVES, KSW>;

  if (smem > 64 * 1024)
    SW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)smem));
  kern<<<grid, WAVES * 64, smem, stream>>>(
      static_cast<uint16_t*>(out), static_cast<const uint16_t*>(q), static_cast<const uint16_t*>(k),
      static_cast<const uint16_t*>(v), seq_starts, seq_lens, num_heads, num_kv_heads, scale, alibi, q_stride,
      k_stride, v_stride, out_stride, sliding_window, num_seqs, max_seq_len, cx);
  SW_CHECK_LAUNCH();
  if (parts > 1) {
    const unsigned col_groups = (unsigned)((num_heads * (HEAD / 4) + 255) / 256);
    context_attention_reduce_kernel<act_t><<<dim3((unsigned)max_seq_len, col_groups, (unsigned)num_seqs), 256, 0, stream>>>(
        static_cast<uint16_t*>(out), cx.part_o, cx.part_m, cx.part_l, seq_starts, seq_lens, num_heads, HEAD, parts,
        cx.total_rows, out_stride);
    SW_CHECK_LAUNCH();
  }
  return SW_OK;
}

// 4 waves (128 query rows) per workgroup, two workgroups per CU.  Without a cached context, prompts of
// kPfWideFrom tokens or more whose 256-row tiles still make kPfWideWorkgroups workgroups (1.5 per CU) take 8 waves
// (256 query rows) that share every K / V stage: half the staging traffic per score.  Shorter prompts and
// smaller launches are faster with the narrow workgroups, which leave fewer CUs idle in the last round.
constexpr int kPfWideFrom = 2048;
constexpr int kPfWideWorkgroups = 384;
// Launches that do not even give every CU one workgroup of 128 rows (one short prompt) take waves in pairs (64 rows per
// workgroup, each wave one of a stage's two score tiles): 1 x 772 tokens 25.9 -> 22.5 us, 772 new on 772 cached 71 -> 65 us.
// From a few hundred workgroups on the pairs lose (3 x 772: 37.6 -> 46.7 us): a wave with one score tile per stage has
// one chain of matrix products to hide its softmax behind instead of two, and every stage is staged for half the rows.
constexpr int kPfPairsUpTo = 256;

template <typename act_t, int HEAD, bool CTX>
int launch_prefill(void* out, const void* q, const void* k, const void* v, const int32_t* seq_starts,
                   const int32_t* seq_lens, int num_seqs, int max_seq_len, int num_heads, int num_kv_heads,
                   float scale, const float* alibi, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                   int64_t out_stride, int sliding_window, const PfContext& cx, hipStream_t stream) {
  int wide = !CTX && HEAD <= 128 && max_seq_len >= kPfWideFrom &&
             (int64_t)((max_seq_len + 8 * kPfRows - 1) / (8 * kPfRows)) * num_heads * num_seqs >= kPfWideWorkgroups;
  // Launches of few workgroups wait for their longest key loop: waves in pairs that split every stage's keys
  const int64_t narrow_wgs = (int64_t)((max_seq_len + kPfWaves * kPfRows - 1) / (kPfWaves * kPfRows)) * num_heads * num_seqs;
  int pairs = pf_tiles(HEAD) == 2 && narrow_wgs <= kPfPairsUpTo && (!CTX || cx.num_parts <= 1);
#ifdef SW_PREFILL_EXPERIMENT
  if (const char* env = getenv("SW_PREFILL_WIDE")) wide = !CTX && HEAD <= 128 && atoi(env) != 0;
  if (const char* env = getenv("SW_PREFILL_PAIRS")) pairs = pf_tiles(HEAD) == 2 && atoi(env) != 0;
#endif
  if constexpr (!CTX && HEAD <= 128) {
    if (wide)
      return launch_prefill_w<act_t, HEAD, CTX, 8, 1>(out, q, k, v, seq_starts, seq_lens, num_seqs, max_seq_len, num_heads,
                                                      num_kv_heads, scale, alibi, q_stride, k_stride, v_stride, out_stride,
                                                      sliding_window, cx, stream);
  }
  if constexpr (pf_tiles(HEAD) == 2) {
    if (pairs)
      return launch_prefill_w<act_t, HEAD, CTX, kPfWaves, 2>(out, q, k, v, seq_starts, seq_lens, num_seqs, max_seq_len,
                                                             num_heads, num_kv_heads, scale, alibi, q_stride, k_stride,
                                                             v_stride, out_stride, sliding_window, cx, stream);
  }
  return launch_prefill_w<act_t, HEAD, CTX, kPfWaves, 1>(out, q, k, v, seq_starts, seq_lens, num_seqs, max_seq_len,
                                                         num_heads, num_kv_heads, scale, alibi, q_stride, k_stride,
                                                         v_stride, out_stride, sliding_window, cx, stream);
}

int prefill_entry(void* out, const void* query, const void* key, const void* value, const int32_t* seq_starts,
                  const int32_t* seq_lens, int num_seqs, int max_seq_len, int num_heads, int num_kv_heads,
                  int head_size, float scale, const float* alibi_slopes, int64_t q_stride, int64_t k_stride,
                  int64_t v_stride, int64_t out_stride, int sliding_window, int act_dtype, const PfContext* cx,
                  void* stream) {
  if (num_seqs == 0 || max_seq_len == 0) return SW_OK;
  SW_CHECK_ARG(num_seqs > 0 && max_seq_len > 0 && num_heads > 0 && num_kv_heads > 0 &&
                   num_heads % num_kv_heads == 0, "prefill_attention: bad shape");
  SW_CHECK_ARG(act_dtype == SW_F16 || act_dtype == SW_BF16,
               "Unsupported data type for prefill attention: %d (float16 / bfloat16 only)", act_dtype);
  SW_CHECK_ARG(num_seqs <= 65535 && num_heads <= 65535, "prefill_attention: grid too large");
  SW_CHECK_ARG((int64_t)max_seq_len * std::max(k_stride, v_stride) * 2 < ((int64_t)1 << 32),
               "prefill_attention: a sequence's keys / values span %lld bytes, more than 32-bit offsets address",
               (long long)max_seq_len * (long long)std::max(k_stride, v_stride) * 2);
  SW_CHECK_ARG(q_stride % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0 && out_stride % 4 == 0 &&
                   reinterpret_cast<uintptr_t>(query) % 16 == 0 && reinterpret_cast<uintptr_t>(key) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(value) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0,
               "prefill_attention: rows must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const PfContext none = {};
#define SW_PF_T(T, HEAD)                                                                                        \
  return cx ? launch_prefill<T, HEAD, true>(out, query, key, value, seq_starts, seq_lens, num_seqs, max_seq_len, \
                                            num_heads, num_kv_heads, scale, alibi_slopes, q_stride, k_stride,   \
                                            v_stride, out_stride, sliding_window, *cx, s)                       \
            : launch_prefill<T, HEAD, false>(out, query, key, value, seq_starts, seq_lens, num_seqs,            \
                                             max_seq_len, num_heads, num_kv_heads, scale, alibi_slopes,         \
                                             q_stride, k_stride, v_stride, out_stride, sliding_window, none, s)
#define SW_PF(HEAD)                        \
  if (act_dtype == SW_BF16) {              \
    SW_PF_T(bf16_t, HEAD);                 \
  } else {                                 \
    SW_PF_T(f16_t, HEAD);                  \
  }
  switch (head_size) {
    case 64: SW_PF(64)
    case 80: SW_PF(80)
    case 96: SW_PF(96)
    case 112: SW_PF(112)
    case 128: SW_PF(128)
    case 256: SW_PF(256)
    default: break;
  }
#undef SW_PF
#undef SW_PF_T
  set_error("Unsupported head size: %d", head_size);
  return SW_ERR_INVALID;
}

}  // namespace sw

extern "C" int sw_prefill_attention(void* out, const void* query, const void* key, const void* value,
                                    const int32_t* seq_starts, const int32_t* seq_lens, int num_seqs,
                                    int max_seq_len, int num_heads, int num_kv_heads, int head_size,
                                    float scale, const float* alibi_slopes, int64_t q_stride,
                                    int64_t k_stride, int64_t v_stride, int64_t out_stride,
                                    int sliding_window, int act_dtype, void* stream) {
  return sw::prefill_entry(out, query, key, value, seq_starts, seq_lens, num_seqs, max_seq_len, num_heads,
                           num_kv_heads, head_size, scale, alibi_slopes, q_stride, k_stride, v_stride, out_stride,
                           sliding_window, act_dtype, nullptr, stream);
}

static int context_attention_entry(void* out, const void* query, const void* key, const void* value,
                                   const void* key_cache, const void* value_cache, const int32_t* block_tables,
                                   int max_blocks_per_seq, const int32_t* seq_starts, const int32_t* new_lens,
                                   const int32_t* ctx_lens, int num_seqs, int max_new_len, int num_heads,
                                   int num_kv_heads, int head_size, int block_size, float scale,
                                   const float* alibi_slopes, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                   int64_t out_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                   int sliding_window, int act_dtype, int64_t total_rows, int num_parts,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace sw;
  SW_CHECK_ARG(key_cache && value_cache && block_tables && ctx_lens && max_blocks_per_seq > 0,
               "context_attention: NULL cache, block table or context lengths");
  SW_CHECK_ARG(block_size > 0 && block_size % 4 == 0, "Unsupported block size: %d", block_size);
  SW_CHECK_ARG(head_size % 8 == 0 && kv_block_stride % 8 == 0 && kv_head_stride % 8 == 0 &&
                   reinterpret_cast<uintptr_t>(key_cache) % 16 == 0 && reinterpret_cast<uintptr_t>(value_cache) % 16 == 0,
               "context_attention: the caches must be 16-byte aligned with strides in multiples of 8 elements");
  PfContext cx = {};
  cx.k_cache = static_cast<const uint16_t*>(key_cache);
  cx.v_cache = static_cast<const uint16_t*>(value_cache);
  cx.block_tables = block_tables;
  cx.ctx_lens = ctx_lens;
  cx.max_blocks = max_blocks_per_seq;
  cx.block_size = block_size;
  cx.k_block_stride = cx.v_block_stride = kv_block_stride;
  cx.k_head_stride = cx.v_head_stride = kv_head_stride;
  cx.num_parts = 1;
  if (num_parts > 1) {
    SW_CHECK_ARG(num_parts <= 64 && total_rows > 0 && workspace != nullptr &&
                     reinterpret_cast<uintptr_t>(workspace) % 16 == 0 &&
                     workspace_bytes >= sw_context_attention_split_workspace_bytes(num_parts, total_rows, num_heads, head_size),
                 "context_attention_split: %d parts of %lld rows need a 16-byte aligned workspace of %lld bytes",
                 num_parts, (long long)total_rows,
                 (long long)sw_context_attention_split_workspace_bytes(num_parts, total_rows, num_heads, head_size));
    const int64_t rows = (int64_t)num_parts * total_rows * num_heads;
    cx.num_parts = num_parts;
    cx.total_rows = total_rows;
    cx.part_o = static_cast<float*>(workspace);
    cx.part_m = cx.part_o + rows * head_size;
    cx.part_l = cx.part_m + rows;
  }
  return prefill_entry(out, query, key, value, seq_starts, new_lens, num_seqs, max_new_len, num_heads, num_kv_heads,
                       head_size, scale, alibi_slopes, q_stride, k_stride, v_stride, out_stride, sliding_window,
                       act_dtype, &cx, stream);
}

extern "C" int64_t sw_context_attention_split_workspace_bytes(int num_parts, int64_t total_rows, int num_heads,
                                                             int head_size) {
  if (num_parts <= 1) return 0;
  return (int64_t)num_parts * total_rows * num_heads * (head_size + 2) * 4;
}

// How many workgroups should share the key stages of one (query tile, sequence, head): 1 (no split) when the launch
// fills the chip by itself, the context is short, or the shapes need the unsplit form.
extern "C" int sw_context_attention_split_parts(int num_seqs, int max_new_len, int num_heads, int head_size,
                                                int block_size, int max_context_len) {
  if (num_seqs <= 0 || max_new_len <= 0 || head_size % 32 != 0 || block_size % 8 != 0) return 1;
  const int64_t workgroups = (int64_t)((max_new_len + 127) / 128) * num_heads * num_seqs;
  // measured on 1 x 512-on-3584 (128 workgroups: 184 us unsplit, 91 us in 4 parts), 2 x 512-on-3584 (256: 185 ->
  // 133 us in 2), 128-on-8192 (32: 369 -> 89 us in 8); launches of 512 workgroups are best left alone
  if (workgroups >= 384 || max_context_len < 1024) return 1;
  int parts = (int)std::min<int64_t>(8, (512 + workgroups / 2) / workgroups);
  parts = std::min(parts, max_context_len / 512);          // at least 8 stages of 64 keys per part
  return std::max(parts, 1);
}

extern "C" int sw_context_attention(void* out, const void* query, const void* key, const void* value,
                                    const void* key_cache, const void* value_cache, const int32_t* block_tables,
                                    int max_blocks_per_seq, const int32_t* seq_starts, const int32_t* new_lens,
                                    const int32_t* ctx_lens, int num_seqs, int max_new_len, int num_heads,
                                    int num_kv_heads, int head_size, int block_size, float scale,
                                    const float* alibi_slopes, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                    int64_t out_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                    int sliding_window, int act_dtype, void* stream) {
  return context_attention_entry(out, query, key, value, key_cache, value_cache, block_tables, max_blocks_per_seq,
                                 seq_starts, new_lens, ctx_lens, num_seqs, max_new_len, num_heads, num_kv_heads,
                                 head_size, block_size, scale, alibi_slopes, q_stride, k_stride, v_stride, out_stride,
                                 kv_block_stride, kv_head_stride, sliding_window, act_dtype, 0, 1, nullptr, 0, stream);
}

extern "C" int sw_context_attention_split(void* out, const void* query, const void* key, const void* value,
                                          const void* key_cache, const void* value_cache, const int32_t* block_tables,
                                          int max_blocks_per_seq, const int32_t* seq_starts, const int32_t* new_lens,
                                          const int32_t* ctx_lens, int num_seqs, int max_new_len, int num_heads,
                                          int num_kv_heads, int head_size, int block_size, float scale,
                                          const float* alibi_slopes, int64_t q_stride, int64_t k_stride,
                                          int64_t v_stride, int64_t out_stride, int64_t kv_block_stride,
                                          int64_t kv_head_stride, int sliding_window, int act_dtype,
                                          int64_t total_rows, int num_parts, void* workspace, int64_t workspace_bytes,
                                          void* stream) {
  return context_attention_entry(out, query, key, value, key_cache, value_cache, block_tables, max_blocks_per_seq,
                                 seq_starts, new_lens, ctx_lens, num_seqs, max_new_len, num_heads, num_kv_heads,
                                 head_size, block_size, scale, alibi_slopes, q_stride, k_stride, v_stride, out_stride,
                                 kv_block_stride, kv_head_stride, sliding_window, act_dtype, total_rows, num_parts,
                                 workspace, workspace_bytes, stream);
}
