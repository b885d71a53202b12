// All-reduce (sum) over the tensor-parallel ranks of one node by direct peer reads.
//
// What it replaces: csrc/custom_all_reduce.cuh:144-329 (start/end barriers, packed_reduce,
// cross_device_reduce_1stage / _2stage) and the dispatch of :486-548, which the reference
// compiles for CUDA only (pybind.cpp:113-127).  Every rank exposes ONE device allocation to its
// peers (HIP IPC; xGMI between GPUs): a signal area, a scratch area for the two-stage result and
// a staging area that holds the rank's input.  Arithmetic follows the reference exactly, so
// every rank gets bitwise the same result:
//   one stage   out[i] = round( f32(in_0[i]) + f32(in_1[i]) + ... + f32(in_{n-1}[i]) )
//   two stages  rank r reduces its 1/n slice in the order r, r+1, ..., r+n-1 (mod n) into its
//               scratch area (reduce-scatter), then every rank copies all slices (all-gather).
//
// Synchronisation: per (workgroup, peer) counters that only grow -- call c uses the values
// 3c+1, 3c+2, 3c+3 -- so nothing is ever reset and a late peer cannot see a stale "ready".
// The call number c lives on the device, one word per workgroup index in the rank's own signal
// area, read at the start of a launch and advanced at its end: a launch carries no per-call
// argument, so it can be captured into a HIP graph and replayed (every rank must still issue the
// same sequence of calls, eager or replayed; workgroup b takes part in the same calls on every
// rank because the launch geometry depends on the message size only).
// A barrier is: wait for my own stores, one relaxed system-scope store of the counter into each
// peer's signal area, spin (bounded) on my own.  Data that crosses ranks never relies on a cache
// write-back: peers' inputs were produced by earlier kernels (complete, hence in memory), and
// the reduce-scatter result is stored write-through; all cross-rank loads bypass the caches.
#include "common.cuh"

#include <algorithm>
#include <cstddef>

namespace sw {
namespace {

constexpr int kArMaxRanks = 8;
constexpr int kArMaxBlocks = 64;
constexpr int kArThreads = 512;
constexpr int kArAuxSystem = 17;                    // sc0 | sc1

struct ArSignal {
  int32_t flag[kArMaxBlocks][kArMaxRanks];          // flag[b][r]: last barrier value rank r announced to workgroup b
  uint32_t calls[kArMaxBlocks];                     // calls[b]: launches workgroup b of THIS rank has completed
  int32_t status;                                   // 1 + peer rank once a bounded wait ran out
  int32_t pad[63];
};
static_assert(sizeof(ArSignal) == 2560, "the signal area's size is part of the ABI");
static_assert(offsetof(ArSignal, status) == sizeof(ArSignal) - 256, "callers read the status word 256 bytes before the end");

struct ArPeers {
  char* base[kArMaxRanks];                          // each rank's exported allocation as mapped here
};

struct ArState {
  ArPeers peers;
  int rank, world;
  int64_t signal_off, scratch_off, staging_off, max_bytes;
};

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

template <typename T> struct Pack;                  // 16 bytes of T <-> 4 or 8 floats
template <> struct Pack<float> {
  static constexpr int N = 4;
  __device__ static void up(u32x4 v, float (&f)[4]) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  __device__ static u32x4 down(const float (&f)[4]) {
    return u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
  }
};
template <> struct Pack<f16_t> {
  static constexpr int N = 8;
  __device__ static void up(u32x4 v, float (&f)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = f16_bits_to_float((uint16_t)(w[i] & 0xffffu));
      f[2 * i + 1] = f16_bits_to_float((uint16_t)(w[i] >> 16));
    }
  }
  __device__ static u32x4 down(const float (&f)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = (uint32_t)float_to_f16_bits(f[2 * i]) | ((uint32_t)float_to_f16_bits(f[2 * i + 1]) << 16);
    return u32x4{w[0], w[1], w[2], w[3]};
  }
};
template <> struct Pack<bf16_t> {
  static constexpr int N = 8;
  __device__ static void up(u32x4 v, float (&f)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static u32x4 down(const float (&f)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = (uint32_t)float_to_bf16_bits(f[2 * i]) | ((uint32_t)float_to_bf16_bits(f[2 * i + 1]) << 16);
    return u32x4{w[0], w[1], w[2], w[3]};
  }
};

// Announces `value` to workgroup blockIdx.x of every peer and waits until every peer has
// announced it to this one.  All earlier stores of the workgroup are complete before the
// announcement leaves.
template <int NR>
__device__ __forceinline__ void ar_barrier(const ArPeers& peers, int64_t signal_off, int rank, int32_t value,
                                           long long max_spins) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < NR) {
    const int peer = threadIdx.x;
    ArSignal* theirs = reinterpret_cast<ArSignal*>(peers.base[peer] + signal_off);
    ArSignal* mine = reinterpret_cast<ArSignal*>(peers.base[rank] + signal_off);
    __hip_atomic_store(&theirs->flag[blockIdx.x][rank], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    long long spins = 0;
    while (__hip_atomic_load(&mine->flag[blockIdx.x][peer], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value < 0) {
      if (++spins > max_spins) {
        __hip_atomic_store(&mine->status, 1 + peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}

// First barrier value of this launch for workgroup blockIdx.x (uniform over the workgroup).
__device__ __forceinline__ int32_t ar_call_value(const ArPeers& peers, int64_t signal_off, int rank) {
  const ArSignal* mine = reinterpret_cast<const ArSignal*>(peers.base[rank] + signal_off);
  const uint32_t c = __hip_atomic_load(&mine->calls[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (int32_t)(3u * c + 1u);
}

// After the launch's last barrier: the next launch of this workgroup index uses the next values.
__device__ __forceinline__ void ar_call_done(const ArPeers& peers, int64_t signal_off, int rank, int32_t value) {
  if (threadIdx.x == 0) {
    ArSignal* mine = reinterpret_cast<ArSignal*>(peers.base[rank] + signal_off);
    __hip_atomic_store(&mine->calls[blockIdx.x], ((uint32_t)value - 1u) / 3u + 1u, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int NR>
struct Rsrcs {
  __amdgpu_buffer_rsrc_t r[NR];
};

// 16-byte vector `idx` of the buffers in `src`, summed in the order src[0], src[1], ...
template <typename T, int NR>
__device__ __forceinline__ u32x4 packed_sum(const Rsrcs<NR>& src, int idx) {
  constexpr int N = Pack<T>::N;
  u32x4 v[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(src.r[i], idx * 16, 0, kArAuxSystem);
  float acc[N];
  Pack<T>::up(v[0], acc);
#pragma unroll
  for (int i = 1; i < NR; ++i) {
    float f[N];
    Pack<T>::up(v[i], f);
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] += f[j];
  }
  return Pack<T>::down(acc);
}

template <typename T, int NR>
__global__ void __launch_bounds__(kArThreads)
all_reduce_1stage_kernel(ArPeers peers, int64_t signal_off, int64_t staging_off, u32x4* __restrict__ out, int rank,
                         int n_vec, long long max_spins) {
  const int32_t value = ar_call_value(peers, signal_off, rank);
  Rsrcs<NR> src;
#pragma unroll
  for (int i = 0; i < NR; ++i)
    src.r[i] = __builtin_amdgcn_make_buffer_rsrc(peers.base[i] + staging_off, 0, n_vec * 16, 0x00020000);
  ar_barrier<NR>(peers, signal_off, rank, value, max_spins);           // every input is in place
  for (int idx = blockIdx.x * kArThreads + threadIdx.x; idx < n_vec; idx += gridDim.x * kArThreads)
    out[idx] = packed_sum<T, NR>(src, idx);
  ar_barrier<NR>(peers, signal_off, rank, value + 1, max_spins);       // nobody still reads my input
  ar_call_done(peers, signal_off, rank, value);
}

template <typename T, int NR>
__global__ void __launch_bounds__(kArThreads)
all_reduce_2stage_kernel(ArPeers peers, int64_t signal_off, int64_t scratch_off, int64_t staging_off,
                         u32x4* __restrict__ out, int rank, int n_vec, long long max_spins) {
  const int32_t value = ar_call_value(peers, signal_off, rank);
  const int part = n_vec / NR;
  const int start = rank * part;
  const int end = rank == NR - 1 ? n_vec : start + part;
  const int largest = n_vec - (NR - 1) * part;
  Rsrcs<NR> src, tmp;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int target = (rank + i) % NR;
    src.r[i] = __builtin_amdgcn_make_buffer_rsrc(peers.base[target] + staging_off, 0, n_vec * 16, 0x00020000);
    tmp.r[i] = __builtin_amdgcn_make_buffer_rsrc(peers.base[target] + scratch_off, 0, largest * 16, 0x00020000);
  }
  const int tid = blockIdx.x * kArThreads + threadIdx.x, stride = gridDim.x * kArThreads;
  ar_barrier<NR>(peers, signal_off, rank, value, max_spins);
  // stage 1, reduce-scatter: my slice, summed starting from my own input, into my scratch area
  for (int idx = start + tid; idx < end; idx += stride)
    __builtin_amdgcn_raw_buffer_store_b128(packed_sum<T, NR>(src, idx), tmp.r[0], (idx - start) * 16, 0, kArAuxSystem);
  ar_barrier<NR>(peers, signal_off, rank, value + 1, max_spins);       // all slices are in memory
  // stage 2, all-gather
  for (int idx = tid; idx < largest; idx += stride) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int target = (rank + i) % NR;
      if (idx < part || target == NR - 1)
        out[target * part + idx] = __builtin_amdgcn_raw_buffer_load_b128(tmp.r[i], idx * 16, 0, kArAuxSystem);
    }
  }
  ar_barrier<NR>(peers, signal_off, rank, value + 2, max_spins);       // nobody still reads my scratch or input
  ar_call_done(peers, signal_off, rank, value);
}

template <typename T, int NR>
int launch_all_reduce(ArState* st, void* out, int n_vec, bool two_stage, hipStream_t stream) {
  const int blocks = std::min(36, (n_vec + kArThreads - 1) / kArThreads);
  const long long max_spins = 20LL * 1000 * 1000;                     // a lost peer ends the wait after seconds
  if (two_stage)
    all_reduce_2stage_kernel<T, NR><<<blocks, kArThreads, 0, stream>>>(st->peers, st->signal_off, st->scratch_off,
                                                                      st->staging_off, static_cast<u32x4*>(out),
                                                                      st->rank, n_vec, max_spins);
  else
    all_reduce_1stage_kernel<T, NR><<<blocks, kArThreads, 0, stream>>>(st->peers, st->signal_off, st->staging_off,
                                                                      static_cast<u32x4*>(out), st->rank, n_vec,
                                                                      max_spins);
  SW_CHECK_LAUNCH();
  return SW_OK;
}

template <typename T>
int dispatch_world(ArState* st, void* out, int n_vec, bool two_stage, hipStream_t stream) {
  switch (st->world) {
    case 2: return launch_all_reduce<T, 2>(st, out, n_vec, two_stage, stream);
    case 4: return launch_all_reduce<T, 4>(st, out, n_vec, two_stage, stream);
    case 6: return launch_all_reduce<T, 6>(st, out, n_vec, two_stage, stream);
    case 8: return launch_all_reduce<T, 8>(st, out, n_vec, two_stage, stream);
    default: break;
  }
  set_error("custom all-reduce only supports 2, 4, 6 or 8 ranks (got %d)", st->world);
  return SW_ERR_INVALID;
}

}  // namespace
}  // namespace sw

using namespace sw;

extern "C" {

int64_t sw_custom_ar_signal_bytes(void) { return (int64_t)sizeof(ArSignal); }

int sw_custom_ar_init(void** peer_bases, int rank, int world, int64_t signal_off, int64_t scratch_off,
                      int64_t staging_off, int64_t max_bytes, void** state_out) {
  SW_CHECK_ARG(peer_bases && state_out, "custom_ar_init: NULL argument");
  SW_CHECK_ARG(world == 2 || world == 4 || world == 6 || world == 8,
               "custom all-reduce only supports 2, 4, 6 or 8 ranks (got %d)", world);
  SW_CHECK_ARG(rank >= 0 && rank < world, "custom_ar_init: rank %d outside 0..%d", rank, world - 1);
  SW_CHECK_ARG(max_bytes > 0 && max_bytes % 16 == 0 && max_bytes < ((int64_t)1 << 31) && signal_off % 64 == 0 &&
                   scratch_off % 16 == 0 && staging_off % 16 == 0,
               "custom_ar_init: offsets must be aligned and max_bytes a multiple of 16 below 2 GiB");
  ArState* st = new ArState();
  for (int i = 0; i < world; ++i) {
    if (!peer_bases[i] || reinterpret_cast<uintptr_t>(peer_bases[i]) % 16 != 0) {
      delete st;
      set_error("custom_ar_init: peer %d has no (16-byte aligned) mapping", i);
      return SW_ERR_INVALID;
    }
    st->peers.base[i] = static_cast<char*>(peer_bases[i]);
  }
  st->rank = rank;
  st->world = world;
  st->signal_off = signal_off;
  st->scratch_off = scratch_off;
  st->staging_off = staging_off;
  st->max_bytes = max_bytes;
  *state_out = st;
  return SW_OK;
}

int sw_custom_all_reduce(void* state, const void* input, void* output, int64_t nbytes, int act_dtype,
                         int algorithm, void* stream) {
  SW_CHECK_ARG(state && output, "custom_all_reduce: NULL argument");
  ArState* st = static_cast<ArState*>(state);
  SW_CHECK_ARG(nbytes > 0 && nbytes % 16 == 0 && nbytes <= st->max_bytes,
               "custom_all_reduce: %lld bytes (must be a multiple of 16, at most %lld)", (long long)nbytes,
               (long long)st->max_bytes);
  SW_CHECK_ARG(reinterpret_cast<uintptr_t>(output) % 16 == 0, "custom_all_reduce: output must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* staging = st->peers.base[st->rank] + st->staging_off;
  if (input != nullptr && input != staging)          // an input that peers cannot read goes through the staging area
    SW_CHECK_HIP(hipMemcpyAsync(staging, input, (size_t)nbytes, hipMemcpyDeviceToDevice, s));
  const int n_vec = (int)(nbytes / 16);
  // the reference's rule (custom_all_reduce.cuh:521-527): two ranks always read directly; more
  // ranks switch to reduce-scatter + all-gather once the message is large
  bool two_stage = st->world > 2 && ((st->world <= 4 && nbytes >= 512 * 1024) || (st->world > 4 && nbytes >= 256 * 1024));
  if (algorithm == 1) two_stage = false;
  if (algorithm == 2) two_stage = st->world > 1;
  if (n_vec < st->world) two_stage = false;
  switch (act_dtype) {
    case SW_F32: return dispatch_world<float>(st, output, n_vec, two_stage, s);
    case SW_F16: return dispatch_world<f16_t>(st, output, n_vec, two_stage, s);
    case SW_BF16: return dispatch_world<bf16_t>(st, output, n_vec, two_stage, s);
    default: break;
  }
  set_error("custom_all_reduce: unsupported data type %d", act_dtype);
  return SW_ERR_INVALID;
}

int sw_custom_ar_dispose(void* state) {
  delete static_cast<ArState*>(state);
  return SW_OK;
}

}  // extern "C"
